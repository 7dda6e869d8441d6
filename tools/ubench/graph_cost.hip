// graph_cost.hip — what a hipGraph of a 16-step chunk (64 short dependent kernels) costs to capture, instantiate and
// replay on this stack, against 64 plain launches: decides whether re-capturing after every neighbour rebuild pays.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(float* x, int n, float a) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * a + 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int n = 1 << 16, K = 64;
    float* x; hipMalloc(&x, n * 4); hipMemset(x, 0, n * 4);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int w = 0; w < 3; ++w) { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, st, x, n, 0.5f); hipStreamSynchronize(st); }
    double t0 = now();
    for (int r = 0; r < 20; ++r) { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, st, x, n, 0.5f); hipStreamSynchronize(st); }
    const double plain = (now() - t0) / 20;
    double cap = 0, inst = 0, rep = 0, upd = 0;
    hipGraphExec_t ex = nullptr;
    for (int r = 0; r < 10; ++r) {
        hipGraph_t g;
        double a = now();
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, st, x, n, 0.5f + 0.001f * r);
        hipStreamEndCapture(st, &g);
        double b = now();
        if (!ex) { hipGraphInstantiate(&ex, g, nullptr, nullptr, 0); inst += now() - b; }
        else { hipGraphNode_t en = nullptr; hipGraphExecUpdateResult ur; double c = now(); hipError_t e = hipGraphExecUpdate(ex, g, &en, &ur); upd += now() - c; if (e != hipSuccess) std::printf("update failed %d\n", (int)e); }
        cap += b - a;
        double c = now();
        for (int q = 0; q < 5; ++q) hipGraphLaunch(ex, st);
        hipStreamSynchronize(st);
        rep += (now() - c) / 5;
        hipGraphDestroy(g);
    }
    std::printf("64 dependent tiny kernels: plain launches + sync %.1f us | capture %.1f us  instantiate (once) %.1f us  exec-update %.1f us  replay + sync %.1f us\n",
                plain, cap / 10, inst, upd / 9, rep / 10);
    return 0;
}
