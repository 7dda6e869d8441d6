// ref_cuda_host.hip — host launchers around the REFERENCE's own pair kernels.
//
// TEST INFRASTRUCTURE ONLY (same rule as mdx_oracle.c).  The only native code the reference holds on this path is
// /root/reference/src/cuda/cuda.cu + util.cu: `coulomb_force_kernel`, `lj_force_kernel`, `lj_V_kernel` (all-pairs,
// one target per thread; unused by the application) and the device functions `lj_force`, `lj_force_v2`,
// `lj_V`, `coulomb_force`, `min_image` they call.  They are plain CUDA C++ with no CUDA-only header, and hipcc - a C++
// compiler of this image - compiles the files AS THEY LIE under /root/reference (nothing is copied, translated or
// stubbed: the kernels below are the reference's object code for gfx950).  oracle/Makefile builds this file into
// oracle/_ref/libref_cuda.so when /root/reference is present; the .so travels to the GPU box, the sources do not.
// tests/test_gpu_reference_kernels.py runs the reference's kernels on the MI355X and pins the oracle's (and the
// engine's) LJ 12-6 force / energy form, `tgt - src` direction and Coulomb form against them.
#include <hip/hip_runtime.h>
#include "cuda.cu"            // -I /root/reference/src/cuda : the reference's file, unmodified

#define RTRY(x) do { if ((x) != hipSuccess) return -1; } while (0)

namespace {
template <typename T> struct DevBuf {
    T* p = nullptr;
    int alloc(size_t n, const T* host) {
        RTRY(hipMalloc((void**)&p, sizeof(T) * (n ? n : 1)));
        if (host && n) RTRY(hipMemcpy(p, host, sizeof(T) * n, hipMemcpyHostToDevice));
        return 0;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

// force on every target from every source: out[3 n_tgt] (zero-initialised here; the kernel accumulates)
extern "C" int ref_lj_force(const float* tgt, size_t n_tgt, const float* src, size_t n_src, const float* sigma /* [n_tgt*n_src] */,
                            const float* eps, float* out) {
    DevBuf<float3> dt, ds, dout; DevBuf<float> dsig, deps;
    if (dt.alloc(n_tgt, (const float3*)tgt) || ds.alloc(n_src, (const float3*)src) || dout.alloc(n_tgt, nullptr) ||
        dsig.alloc(n_tgt * n_src, sigma) || deps.alloc(n_tgt * n_src, eps)) return -1;
    RTRY(hipMemset(dout.p, 0, sizeof(float3) * n_tgt));
    hipLaunchKernelGGL(lj_force_kernel, dim3((unsigned)((n_tgt + 63) / 64)), dim3(64), 0, 0, dout.p, ds.p, dt.p, dsig.p, deps.p, n_src, n_tgt);
    RTRY(hipDeviceSynchronize());
    RTRY(hipMemcpy(out, dout.p, sizeof(float3) * n_tgt, hipMemcpyDeviceToHost));
    return 0;
}

// charges[max(n_tgt, n_src)]: the kernel reads the charge of source i and of target j from the same array
extern "C" int ref_coulomb_force(const float* tgt, size_t n_tgt, const float* src, size_t n_src, const float* charges, size_t n_q, float* out) {
    if (n_q < n_tgt || n_q < n_src) return -2;
    DevBuf<float3> dt, ds, dout; DevBuf<float> dq;
    if (dt.alloc(n_tgt, (const float3*)tgt) || ds.alloc(n_src, (const float3*)src) || dout.alloc(n_tgt, nullptr) || dq.alloc(n_q, charges)) return -1;
    RTRY(hipMemset(dout.p, 0, sizeof(float3) * n_tgt));
    hipLaunchKernelGGL(coulomb_force_kernel, dim3((unsigned)((n_tgt + 63) / 64)), dim3(64), 0, 0, dout.p, ds.p, dt.p, dq.p, n_src, n_tgt);
    RTRY(hipDeviceSynchronize());
    RTRY(hipMemcpy(out, dout.p, sizeof(float3) * n_tgt, hipMemcpyDeviceToHost));
    return 0;
}

// LJ energy of every target with every source (the kernel uses sigmas[0], epsilons[0] for all pairs)
extern "C" int ref_lj_V(const float* p0_src, size_t n_src, const float* p1_tgt, size_t n_tgt, float sigma, float eps, float* out /* [n_tgt] */) {
    DevBuf<float3> d0, d1; DevBuf<float> dout, dsig, deps;
    if (d0.alloc(n_src, (const float3*)p0_src) || d1.alloc(n_tgt, (const float3*)p1_tgt) || dout.alloc(n_tgt, nullptr) ||
        dsig.alloc(1, &sigma) || deps.alloc(1, &eps)) return -1;
    RTRY(hipMemset(dout.p, 0, sizeof(float) * n_tgt));
    hipLaunchKernelGGL(lj_V_kernel, dim3((unsigned)((n_tgt + 63) / 64)), dim3(64), 0, 0, dout.p, d0.p, d1.p, dsig.p, deps.p, n_src, n_tgt);
    RTRY(hipDeviceSynchronize());
    RTRY(hipMemcpy(out, dout.p, sizeof(float) * n_tgt, hipMemcpyDeviceToHost));
    return 0;
}

// the reference's minimum image (util.cu:65-71) for one difference vector
__global__ void ref_min_image_kernel(float3 ext, float3 dv, float3* out) { *out = min_image(ext, dv); }
extern "C" int ref_min_image(const float ext[3], const float dv[3], float out[3]) {
    DevBuf<float3> d;
    if (d.alloc(1, nullptr)) return -1;
    hipLaunchKernelGGL(ref_min_image_kernel, dim3(1), dim3(1), 0, 0, make_float3(ext[0], ext[1], ext[2]), make_float3(dv[0], dv[1], dv[2]), d.p);
    RTRY(hipDeviceSynchronize());
    RTRY(hipMemcpy(out, d.p, sizeof(float3), hipMemcpyDeviceToHost));
    return 0;
}
