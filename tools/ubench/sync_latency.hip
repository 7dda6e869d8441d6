// How long does the host wait for a few device words?  Three ways of getting 8 words of a finished kernel to the host:
//   A  hipMemcpyAsync into pageable memory + hipStreamSynchronize        (what the list rebuild did until round 2)
//   B  a one-thread kernel writing pinned host memory + hipStreamSynchronize
//   C  the same kernel writing a sequence word last, the host spinning on it (no runtime call on the wait path)
// Build: hipcc --offload-arch=gfx950 -O2 -o sync_latency sync_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(uint32_t* d, uint32_t v) { d[threadIdx.x & 7] = v + threadIdx.x; }
__global__ void rb(const uint32_t* d, volatile uint32_t* out, uint32_t seq) {
    for (int i = 0; i < 8; ++i) out[i] = d[i];
    __threadfence_system();
    out[15] = seq;
}
int main() {
    uint32_t *d = nullptr, *pin = nullptr;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipMalloc(&d, 64));
    CK(hipHostMalloc((void**)&pin, 64, hipHostMallocDefault));
    pin[15] = 0;
    uint32_t pageable[8];
    const int n = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        double best = 1e9, sum = 0;
        for (int it = 0; it < n + 100; ++it) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, d, (uint32_t)it);
            if (mode == 0) {
                CK(hipMemcpyAsync(pageable, d, 32, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            } else if (mode == 1) {
                hipLaunchKernelGGL(rb, dim3(1), dim3(1), 0, st, d, pin, (uint32_t)it + 1);
                CK(hipStreamSynchronize(st));
            } else {
                hipLaunchKernelGGL(rb, dim3(1), dim3(1), 0, st, d, pin, (uint32_t)(it + 1 + 100000));
                volatile uint32_t* seq = pin + 15;
                while (*seq != (uint32_t)(it + 1 + 100000)) {}
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (it >= 100) { sum += us; if (us < best) best = us; }
        }
        CK(hipStreamSynchronize(st));
        printf("mode %c: launch + readback + wait  mean %.1f us  best %.1f us\n", 'A' + mode, sum / n, best);
    }
    return 0;
}
