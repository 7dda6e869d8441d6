// Can a memory-bound chain run BESIDE a VALU-bound kernel on compute units of its own?  The reciprocal-space chain of SPME (spread,
// FFT passes, solve, gather: HBM-bound, ~0.4 ms alone at 1 M sites) runs on a side stream next to the pair kernel (VALU-bound, 16 k
// one-wave workgroups that refill every freed slot): its first kernel waits 0.36 ms for slots.  hipExtStreamCreateWithCUMask gives a
// stream its own CUs.  This prices it:
//   1  which (XCC, SE, CU) a mask bit selects (HW_ID / XCC_ID of the waves of a kernel launched on a masked stream)
//   2  the streaming bandwidth of a kernel confined to m CUs per XCD (copy of 256 MB)
//   3  a VALU-bound kernel on the complement mask + the copy on its own CUs, started together: both durations against running alone
// Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask cu_mask.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void where_kernel(uint32_t* out) {
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
        const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
        out[blockIdx.x] = (hw & 0xFFFFu) | ((xcc & 0xFu) << 16);
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000ull) {}      // 20 us: the grid spreads over every CU it may use
}
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(64) void valu_kernel(float* out, int iters) {      // one-wave workgroups, like the pair kernel's
    float x = (float)threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f;
    for (int i = 0; i < iters; ++i) { x = __builtin_fmaf(x, y, z); y = __builtin_fmaf(y, 0.99999f, 1e-6f); z = __builtin_fmaf(z, x, -x * z); }
    if (x == 12345.f) out[blockIdx.x] = x + y + z;
}

static std::vector<uint32_t> mask_per_xcd(int first, int count, bool complement) {
    // candidate layout: bit b <-> XCD b % 8, CU b / 8 of that XCD (checked by part 1)
    std::vector<uint32_t> m(8, 0u);
    for (int b = 0; b < 256; ++b) {
        const int cu = b / 8;
        const bool in = cu >= first && cu < first + count;
        if (in != complement) m[b / 32] |= 1u << (b % 32);
    }
    return m;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
    uint32_t* where = nullptr; CK(hipHostMalloc((void**)&where, 4096 * 4, hipHostMallocDefault));
    // ---- 1: what a mask selects
    for (int variant = 0; variant < 4; ++variant) {
        std::vector<uint32_t> m(8, 0u);
        const char* what = "";
        if (variant == 0) { m[0] = 0xFFu; what = "bits 0-7"; }
        if (variant == 1) { m[0] = 0xFFFFFFFFu; what = "bits 0-31"; }
        if (variant == 2) { m[7] = 0xFF000000u; what = "bits 248-255"; }
        if (variant == 3) { m = mask_per_xcd(0, 3, false); what = "bits 0-23 (3 per XCD if bit b <-> XCD b % 8)"; }
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, m.data()) != hipSuccess) { printf("hipExtStreamCreateWithCUMask refused\n"); return 1; }
        unsigned int fl = 99; (void)hipStreamGetFlags(s, &fl);
        hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 0, s, where);
        CK(hipStreamSynchronize(s));
        std::set<uint32_t> cus; int per_xcc[16] = {};
        for (int i = 0; i < 2048; ++i) cus.insert(((where[i] >> 16) << 16) | (where[i] & 0xFF00u));     // xcc | se, sh, cu
        for (uint32_t c : cus) per_xcc[c >> 16]++;
        printf("mask %-48s stream flags %u: %zu distinct CUs; per XCC:", what, fl, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("\n   ");
        int k = 0; for (uint32_t c : cus) { if (k++ < 12) printf(" xcc%u/se%u/cu%u", c >> 16, (c >> 13) & 7u, (c >> 8) & 15u); }
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    // ---- 2 / 3
    const size_t n = 16u << 20;     // 16 M float4 = 256 MB each way
    float4 *a = nullptr, *b = nullptr; float* sink = nullptr;
    CK(hipMalloc((void**)&a, n * 16)); CK(hipMalloc((void**)&b, n * 16)); CK(hipMalloc((void**)&sink, 1 << 20));
    CK(hipMemset(a, 1, n * 16));
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    hipStream_t full; CK(hipStreamCreateWithFlags(&full, hipStreamNonBlocking));
    auto time_on = [&](hipStream_t s, auto&& launch) { float best = 1e9f; for (int r = 0; r < 5; ++r) { hipEventRecord(e0, s); launch(s); hipEventRecord(e1, s); hipStreamSynchronize(s); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); } return best; };
    const int valu_iters = 20000, valu_wgs = 65536;
    const float copy_full = time_on(full, [&](hipStream_t s) { hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, s, a, b, n); });
    const float valu_full = time_on(full, [&](hipStream_t s) { hipLaunchKernelGGL(valu_kernel, dim3(valu_wgs), dim3(64), 0, s, sink, valu_iters); });
    printf("alone on 256 CUs: copy 2 x 256 MB %.1f us (%.2f TB/s), VALU kernel %.1f us\n", copy_full * 1e3, 2.0 * n * 16 / copy_full / 1e9, valu_full * 1e3);
    {   // both unmasked, started together (today's arrangement)
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEventRecord(e0, full); hipEventRecord(f0, s2);
        hipLaunchKernelGGL(valu_kernel, dim3(valu_wgs), dim3(64), 0, full, sink, valu_iters);
        hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, s2, a, b, n);
        hipEventRecord(e1, full); hipEventRecord(f1, s2);
        CK(hipDeviceSynchronize());
        float tv, tc, tot; hipEventElapsedTime(&tv, e0, e1); hipEventElapsedTime(&tc, f0, f1); hipEventElapsedTime(&tot, e0, f1);
        printf("unmasked side by side: VALU %.1f us, copy %.1f us (e0 -> copy end %.1f us)\n", tv * 1e3, tc * 1e3, tot * 1e3);
        CK(hipStreamDestroy(s2));
    }
    for (int m_cus : {1, 2, 3, 4, 6, 8}) {
        std::vector<uint32_t> mm = mask_per_xcd(0, m_cus, false), mc = mask_per_xcd(0, m_cus, true);
        hipStream_t sm, sc;
        CK(hipExtStreamCreateWithCUMask(&sm, 8, mm.data())); CK(hipExtStreamCreateWithCUMask(&sc, 8, mc.data()));
        const int wgs = m_cus * 8 * 8;
        const float copy_alone = time_on(sm, [&](hipStream_t s) { hipLaunchKernelGGL(copy_kernel, dim3(wgs), dim3(256), 0, s, a, b, n); });
        const float valu_alone = time_on(sc, [&](hipStream_t s) { hipLaunchKernelGGL(valu_kernel, dim3(valu_wgs), dim3(64), 0, s, sink, valu_iters); });
        hipEventRecord(e0, sc); hipEventRecord(f0, sm);
        hipLaunchKernelGGL(valu_kernel, dim3(valu_wgs), dim3(64), 0, sc, sink, valu_iters);
        hipLaunchKernelGGL(copy_kernel, dim3(wgs), dim3(256), 0, sm, a, b, n);
        hipEventRecord(e1, sc); hipEventRecord(f1, sm);
        CK(hipDeviceSynchronize());
        float tv, tc; hipEventElapsedTime(&tv, e0, e1); hipEventElapsedTime(&tc, f0, f1);
        printf("%d CUs per XCD (%3d CUs) for the copy: alone %.1f us (%.2f TB/s); VALU kernel on the other %3d CUs alone %.1f us; side by side: VALU %.1f us, copy %.1f us\n",
               m_cus, m_cus * 8, copy_alone * 1e3, 2.0 * n * 16 / copy_alone / 1e9, 256 - m_cus * 8, valu_alone * 1e3, tv * 1e3, tc * 1e3);
        CK(hipStreamDestroy(sm)); CK(hipStreamDestroy(sc));
    }
    return 0;
}
