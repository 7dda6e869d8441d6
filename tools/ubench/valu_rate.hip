// valu_rate.hip — what does one wave64 fp32 VALU instruction cost on gfx950, and how many waves per SIMD does a
// dependent chain need to saturate the pipe?  Settles whether the pair kernel (3.5 cycles per VALU instruction per
// SIMD at 4 waves/SIMD) is issue-bound or latency-bound.
//   ILP = independent FMA chains per wave (1 = fully dependent), W = waves per SIMD (blocks of 64 threads, 4*W per CU)
// Output: cycles per VALU instruction per SIMD = elapsed * 2.4e9 / (instructions issued on one SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(64) void fma_chain(float* out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) x[k] = (float)threadIdx.x + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64 / ILP; ++r)
#pragma unroll
            for (int k = 0; k < ILP; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s += x[k];
    if (s == 12345.678f) out[0] = s;
}

// the pair kernel's body shape: dependent chain with a transcendental and an exec-mask round trip
__global__ __launch_bounds__(64) void chain_branchy(float* out, int iters, float a, float rc2) {
    float x = (float)threadIdx.x * 0.01f + 1.0f, acc = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float r2 = x * x + a;
            if (r2 < rc2) {           // exec-masked region (all lanes pass: measures the branch machinery)
                float ri = __builtin_amdgcn_rsqf(r2);
                float t = ri * ri; t = t * t * t; acc += t * (2.f * t - 1.f) * ri;
            }
            x += 1e-6f;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// does the SIMD skip the passes of a wave64 instruction whose lanes are all masked off?  `keep` selects the active lanes
__global__ __launch_bounds__(64) void fma_masked(float* out, int iters, float a, float b, unsigned long long keep) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (float)threadIdx.x + k;
    if ((keep >> threadIdx.x) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    if (s == 12345.678f) out[0] = s;
}

template <typename K, typename... A>
static double run(K k, int blocks, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    float* out; hipMalloc(&out, 4);
    const int iters = 4000;
    const double clk = 2.4e9;
    printf("independent/dependent v_fma_f32: cycles per instruction per SIMD (256 CUs x 4 SIMDs)\n");
    for (int W : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * W;
        const double n_simd = (double)iters * 64 * W;   // instructions issued on one SIMD
        double t1 = run(fma_chain<1>, blocks, out, iters, 1.0001f, 0.5f);
        double t2 = run(fma_chain<2>, blocks, out, iters, 1.0001f, 0.5f);
        double t4 = run(fma_chain<4>, blocks, out, iters, 1.0001f, 0.5f);
        double t16 = run(fma_chain<16>, blocks, out, iters, 1.0001f, 0.5f);
        printf("W=%d waves/SIMD: ILP1 %.2f  ILP2 %.2f  ILP4 %.2f  ILP16 %.2f\n", W, t1 * clk / n_simd, t2 * clk / n_simd,
               t4 * clk / n_simd, t16 * clk / n_simd);
    }
    printf("branchy pair-like chain (13 VALU + exec round trip per unit): cycles per unit per SIMD\n");
    for (int W : {1, 2, 4, 5, 6, 8}) {
        const int blocks = 256 * 4 * W;
        double t = run(chain_branchy, blocks, out, iters, 0.25f, 1e30f);
        printf("W=%d: %.1f cycles per unit per SIMD, %.1f per unit per wave\n", W, t * clk / ((double)iters * 8 * W), t * clk / ((double)iters * 8));
    }
    printf("exec-masked v_fma_f32 (8 waves/SIMD, ILP 8): cycles per instruction per SIMD by active-lane pattern\n");
    struct { const char* name; unsigned long long keep; } pats[] = {
        {"all 64", ~0ull}, {"lanes 0-31", 0xFFFFFFFFull}, {"lanes 0-15", 0xFFFFull}, {"lanes 0-7", 0xFFull},
        {"lanes 0-15 + 32-47", 0x0000FFFF0000FFFFull}, {"every 2nd lane", 0x5555555555555555ull},
        {"every 8th lane", 0x0101010101010101ull}, {"lane 0 only", 1ull}};
    for (auto& pt : pats) {
        const int W = 8, blocks = 256 * 4 * W;
        double t = run(fma_masked, blocks, out, iters, 1.0001f, 0.5f, pt.keep);
        printf("  %-20s %.2f\n", pt.name, t * clk / ((double)iters * 64 * W));
    }
    return 0;
}
