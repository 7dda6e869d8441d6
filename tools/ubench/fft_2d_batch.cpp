// The two transforms either side of a hand-written x pass: batched 2-D R2C / C2R over (y, z) for every x, half-complex rows padded.
// Build: hipcc -O2 -o fft_2d_batch fft_2d_batch.cpp -lhipfft
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#define CK(x) do { if ((x) != hipSuccess) { printf("%s failed\n", #x); return 1; } } while (0)
int main() {
    for (int K : {200, 256, 96}) {
        const int kh = K / 2 + 1, pitch = (kh + 15) & ~15;
        float* re = nullptr; hipfftComplex* cx = nullptr;
        CK(hipMalloc((void**)&re, sizeof(float) * (size_t)K * K * K));
        CK(hipMalloc((void**)&cx, sizeof(hipfftComplex) * (size_t)K * K * pitch));
        CK(hipMemset(re, 0, sizeof(float) * (size_t)K * K * K));
        int n2[2] = {K, K}, in2[2] = {K, K}, on2[2] = {K, pitch};
        hipfftHandle f, b, f3, b3;
        if (hipfftPlanMany(&f, 2, n2, in2, 1, K * K, on2, 1, K * pitch, HIPFFT_R2C, K) != HIPFFT_SUCCESS ||
            hipfftPlanMany(&b, 2, n2, on2, 1, K * pitch, in2, 1, K * K, HIPFFT_C2R, K) != HIPFFT_SUCCESS) { printf("2-D plan refused\n"); return 1; }
        int n3[3] = {K, K, K}, in3[3] = {K, K, K}, on3[3] = {K, K, pitch};
        hipfftPlanMany(&f3, 3, n3, in3, 1, K * K * K, on3, 1, K * K * pitch, HIPFFT_R2C, 1);
        hipfftPlanMany(&b3, 3, n3, on3, 1, K * K * pitch, in3, 1, K * K * K, HIPFFT_C2R, 1);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int which = 0; which < 2; ++which) {
            for (int w = 0; w < 3; ++w) { hipfftExecR2C(which ? f3 : f, re, cx); hipfftExecC2R(which ? b3 : b, cx, re); }
            CK(hipDeviceSynchronize());
            const int reps = 50;
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) { hipfftExecR2C(which ? f3 : f, re, cx); hipfftExecC2R(which ? b3 : b, cx, re); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("K %3d pitch %3d: %s forward + inverse %.1f us\n", K, pitch, which ? "3-D            " : "2-D x K batches", ms * 1e3 / reps);
        }
    }
    return 0;
}
