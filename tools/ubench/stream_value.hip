// What does it cost to order a kernel on stream B behind a kernel on stream A?  The decomposed step hands work between its compute
// stream and its communication stream four times per step; this prices the ways of doing it on this stack (MI355X, ROCm 7.2):
//   E  same stream: K1 -> K2                                  (the dependent-launch boundary, for scale)
//   A  hipEventRecord(A) + hipStreamWaitEvent(B) -> K2
//   B  K1 stores a device word at its end; hipStreamWaitValue32(B, word, >=) -> K2      (command-processor wait: no CU involved)
//   C  hipStreamWriteValue32(A, word) behind K1; K2 (already resident on B) polls the word (command-processor write)
//   D  K1 stores the word itself; K2 (already resident) polls it                         (in-kernel hand-off, for scale)
// Times are from K1's last instruction to K2's first (device wall clock, 100 MHz), averaged.
// Build: hipcc --offload-arch=gfx950 -O2 -o stream_value stream_value.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k1(unsigned long long* t_end, uint32_t* word, uint32_t v, int busy_us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)busy_us * 100ull) {}
    if (threadIdx.x == 0) {
        *t_end = wall_clock64();
        if (word) { __threadfence_system(); __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}
__global__ void k2(unsigned long long* t_start) { if (threadIdx.x == 0) *t_start = wall_clock64(); }
__global__ void k2_poll(unsigned long long* t_seen, const uint32_t* word, uint32_t v) {
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < v) __builtin_amdgcn_s_sleep(2);
        *t_seen = wall_clock64();
    }
}

int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned long long* t = nullptr;
    CK(hipHostMalloc((void**)&t, 64, hipHostMallocDefault));
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    uint32_t* word_dev = nullptr; uint32_t* word_sig = nullptr; uint32_t* word_host = nullptr;
    CK(hipMalloc((void**)&word_dev, 64));
    CK(hipMemset(word_dev, 0, 64));
    if (hipExtMallocWithFlags((void**)&word_sig, 64, hipMallocSignalMemory) != hipSuccess) { word_sig = nullptr; (void)hipGetLastError(); printf("hipMallocSignalMemory: not available\n"); }
    else CK(hipMemset(word_sig, 0, 64));
    CK(hipHostMalloc((void**)&word_host, 64, hipHostMallocDefault));
    word_host[0] = 0;
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int n = 300;
    uint32_t gen = 0;
    for (int busy : {2, 30}) {
        {   // E
            double sum = 0;
            for (int it = 0; it < n; ++it) {
                hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, sa, t, (uint32_t*)nullptr, 0u, busy);
                hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, sa, t + 1);
                CK(hipStreamSynchronize(sa));
                sum += (double)(t[1] - t[0]) * 0.01;
            }
            printf("busy %2d us  E same stream K1 -> K2:                      %.2f us\n", busy, sum / n);
        }
        {   // A
            double sum = 0;
            for (int it = 0; it < n; ++it) {
                hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, sa, t, (uint32_t*)nullptr, 0u, busy);
                CK(hipEventRecord(ev, sa));
                CK(hipStreamWaitEvent(sb, ev, 0));
                hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, sb, t + 1);
                CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(sa));
                sum += (double)(t[1] - t[0]) * 0.01;
            }
            printf("busy %2d us  A event record + stream wait event:          %.2f us\n", busy, sum / n);
        }
        struct W { const char* name; uint32_t* p; } words[3] = {{"hipMalloc word", word_dev}, {"signal-memory word", word_sig}, {"pinned host word", word_host}};
        for (const W& w : words) {
            if (!w.p) continue;
            double sum = 0; bool ok = true;
            for (int it = 0; it < n && ok; ++it) {
                ++gen;
                hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, sa, t, w.p, gen, busy);
                if (hipStreamWaitValue32(sb, w.p, gen, hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
                hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, sb, t + 1);
                CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(sa));
                sum += (double)(t[1] - t[0]) * 0.01;
            }
            if (ok) printf("busy %2d us  B device store + hipStreamWaitValue32 (%s): %.2f us\n", busy, w.name, sum / n);
            else printf("busy %2d us  B hipStreamWaitValue32 (%s): refused\n", busy, w.name);
        }
        for (const W& w : words) {
            if (!w.p) continue;
            double sum = 0; bool ok = true;
            for (int it = 0; it < n && ok; ++it) {
                ++gen;
                hipLaunchKernelGGL(k2_poll, dim3(1), dim3(64), 0, sb, t + 1, w.p, gen);
                hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, sa, t, (uint32_t*)nullptr, 0u, busy);
                if (hipStreamWriteValue32(sa, w.p, gen, 0) != hipSuccess) { (void)hipGetLastError(); ok = false; }
                CK(hipStreamSynchronize(sa));
                if (!ok) { if (w.p == word_host) w.p[0] = gen; else CK(hipMemcpy(w.p, &gen, 4, hipMemcpyHostToDevice)); }
                CK(hipStreamSynchronize(sb));
                sum += (double)(t[1] - t[0]) * 0.01;
            }
            if (ok) printf("busy %2d us  C hipStreamWriteValue32 (%s) + resident poller: %.2f us\n", busy, w.name, sum / n);
            else printf("busy %2d us  C hipStreamWriteValue32 (%s): refused\n", busy, w.name);
        }
        {   // D
            double sum = 0;
            for (int it = 0; it < n; ++it) {
                ++gen;
                hipLaunchKernelGGL(k2_poll, dim3(1), dim3(64), 0, sb, t + 1, word_dev, gen);
                hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, sa, t, word_dev, gen, busy);
                CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
                sum += (double)(t[1] - t[0]) * 0.01;
            }
            printf("busy %2d us  D device store + resident poller:             %.2f us\n", busy, sum / n);
        }
    }
    return 0;
}
