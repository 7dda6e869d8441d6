// What does a device-wide barrier cost on MI355X (8 XCDs, no shared L2)?  A persistent kernel that ran a whole chunk of MD steps
// would need two per step in place of two kernel launches (~9 us each in a dependent stream, DESIGN.md section 4 "Small systems").
// G workgroups of 256 threads, all resident; barrier = one device-scope atomic add per workgroup + a spin on a generation word.
// Build: hipcc --offload-arch=gfx950 -O2 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = *gen;
        __threadfence();
        if (atomicAdd(count, 1u) == nwg - 1u) { *count = 0u; __threadfence(); atomicAdd((unsigned*)gen, 1u); }
        else while (*gen == g) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void spin(unsigned* count, unsigned* gen, unsigned nwg, int iters, float* sink) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) { v = v * 1.0001f + 1.0f; grid_barrier(count, gen, nwg); }
    if (v == -1.f) sink[0] = v;
}
int main() {
    unsigned *count, *gen; float* sink;
    CK(hipMalloc(&count, 64)); CK(hipMalloc(&gen, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(count, 0, 64)); CK(hipMemset(gen, 0, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (unsigned nwg : {64u, 256u, 512u, 1024u}) {
        for (int iters : {200, 2000}) {
            void* args[] = {&count, &gen, &nwg, &iters, &sink};
            CK(hipMemset(count, 0, 64));
            CK(hipEventRecord(a));
            CK(hipLaunchCooperativeKernel((const void*)spin, dim3(nwg), dim3(256), args, 0, nullptr));
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (iters == 2000) printf("%4u workgroups: %.2f us per device-wide barrier\n", nwg, 1e3 * ms / iters);
        }
    }
    return 0;
}
