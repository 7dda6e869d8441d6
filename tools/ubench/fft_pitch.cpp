// Does the row pitch of the half-complex mesh matter to rocFFT's strided 200-point passes?  hipfftPlan3d packs the R2C output as
// [K][K][K/2+1] complex (pitch 101 x 8 B = 808 B: no two rows start on the same 64-byte phase); hipfftPlanMany's advanced layout can pad
// it.  Times forward + inverse for K = 200 (and 192, 216, 256 for scale) with pitches K/2+1, and padded to multiples of 4, 8, 16, 32.
// Build: hipcc -O2 -o fft_pitch fft_pitch.cpp -lhipfft ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { printf("%s failed\n", #x); return 1; } } while (0)
int main() {
    for (int K : {200, 192, 216, 256}) {
        for (int padto : {1, 4, 8, 16, 32}) {
            const int kh = K / 2 + 1, pitch = (kh + padto - 1) / padto * padto;
            for (int inplace = 0; inplace < 2; ++inplace) {
                const int rpitch = inplace ? 2 * pitch : K;
                float* re = nullptr; hipfftComplex* cx = nullptr;
                CK(hipMalloc((void**)&cx, sizeof(hipfftComplex) * (size_t)K * K * pitch));
                if (inplace) re = (float*)cx; else CK(hipMalloc((void**)&re, sizeof(float) * (size_t)K * K * rpitch));
                CK(hipMemset(cx, 0, sizeof(hipfftComplex) * (size_t)K * K * pitch));
                int n[3] = {K, K, K}, inembed[3] = {K, K, rpitch}, onembed[3] = {K, K, pitch};
                hipfftHandle f, b;
                if (hipfftPlanMany(&f, 3, n, inembed, 1, K * K * rpitch, onembed, 1, K * K * pitch, HIPFFT_R2C, 1) != HIPFFT_SUCCESS ||
                    hipfftPlanMany(&b, 3, n, onembed, 1, K * K * pitch, inembed, 1, K * K * rpitch, HIPFFT_C2R, 1) != HIPFFT_SUCCESS) { printf("K %d pitch %d: plan refused\n", K, pitch); continue; }
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int w = 0; w < 3; ++w) { hipfftExecR2C(f, re, cx); hipfftExecC2R(b, cx, re); }
                CK(hipDeviceSynchronize());
                const int reps = 50;
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < reps; ++r) { hipfftExecR2C(f, re, cx); hipfftExecC2R(b, cx, re); }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("K %3d  complex pitch %3d (%s): forward + inverse %.1f us\n", K, pitch, inplace ? "in place" : "out of place", ms * 1e3 / reps);
                hipfftDestroy(f); hipfftDestroy(b);
                if (!inplace) (void)hipFree(re);
                (void)hipFree(cx);
            }
        }
    }
    return 0;
}
