// mdx_integrate.hip — velocity-Verlet kick/drift, the rebuild trigger and the kinetic reduction.
//
// `MdState::step` with `Integrator::VerletVelocity` (/root/reference README.md:237,
// src/ui/panels/md.rs:303-305) as kick(dt/2) - drift(dt) - [forces] - kick(dt/2); inside a
// multi-step burst the closing half kick of step n and the opening half kick of step n+1 are one
// pass (mode 1), so a step streams x, v, f once: R posq+vel+force+ref (64 B), W posq+vel (32 B)
// per slot, one slot per lane, 16-B accesses.  Pure HBM streaming.
//
// The same pass measures each atom's squared displacement from its position at the last
// neighbour rebuild; a wave whose maximum exceeds (skin/2)^2 raises ctl.disp2[step+1] with an
// atomicMax (non-negative floats order like their bit patterns).  Downstream kernels gate on
// that word.
#include "mdx_internal.h"

template <int MODE>  // 0: half kick + drift, 1: full kick + drift, 2: closing half kick
__global__ __launch_bounds__(256) void integrate_kernel(uint32_t S, float dt, float4* __restrict__ posq,
                                                        float4* __restrict__ vel, const float4* __restrict__ force,
                                                        const float4* __restrict__ ref, const uint32_t* gate_in,
                                                        uint32_t* disp_out, uint32_t thr_bits) {
    const uint32_t gate = gate_in ? *gate_in : 0u;
    if (gate > thr_bits) {  // list already stale: stay a no-op, keep the flag raised
        if (MODE != 2 && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(disp_out, gate);
        return;
    }
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    float d2 = 0.f;
    if (s < S) {
        float4 v = vel[s];
        if (v.w != 0.f) {   // w = 418.4/m; 0 marks static, ghost and dummy slots
            const float4 f = force[s];
            const float kdt = (MODE == 1 ? dt : 0.5f * dt) * v.w;
            v.x += kdt * f.x; v.y += kdt * f.y; v.z += kdt * f.z;
            vel[s] = v;
            if (MODE != 2) {
                float4 p = posq[s];
                p.x += dt * v.x; p.y += dt * v.y; p.z += dt * v.z;
                posq[s] = p;
                const float4 r = ref[s];
                const float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
                d2 = dx * dx + dy * dy + dz * dz;
                if (!(d2 < 1.0e30f)) d2 = 3.0e38f;  // NaN/inf -> huge, forces a stop
            }
        }
    }
    if (MODE != 2) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
        // only a wave that actually crossed the threshold touches the flag word: in the common
        // (not stale) case the pass issues no atomics at all.  One contended atomicMax per wave
        // cost 9x the streaming time of this kernel (~12 ns each, serialised in L2).
        if ((threadIdx.x & 63) == 0 && __float_as_uint(d2) > thr_bits) atomicMax(disp_out, __float_as_uint(d2));
    }
}

// kinetic energy (kcal/mol) and max |F|^2 over mobile atoms
__global__ __launch_bounds__(256) void kinetic_kernel(uint32_t S, const float4* __restrict__ vel,
                                                      const float4* __restrict__ force, double* energy,
                                                      uint32_t* maxf2_bits) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    double ke = 0.0;
    float f2 = 0.f;
    if (s < S) {
        const float4 v = vel[s];
        if (v.w != 0.f) {
            ke = 0.5 * ((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z) / (double)v.w;
            const float4 f = force[s];
            f2 = f.x * f.x + f.y * f.y + f.z * f.z;
            if (!(f2 < 3.0e38f)) f2 = 3.0e38f;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        ke += __shfl_xor(ke, m);
        f2 = fmaxf(f2, __shfl_xor(f2, m));
    }
    if ((threadIdx.x & 63) == 0) {
        if (ke != 0.0) atomicAdd(&energy[EN_KIN], ke);
        if (f2 > 0.f) atomicMax(maxf2_bits, __float_as_uint(f2));
    }
}

int mdx_launch_integrate(mdx_handle* h, int mode, float dt, const uint32_t* d_gate_in, uint32_t* d_disp_out,
                         uint32_t thr_bits) {
    const dim3 g((h->S + 255) / 256), b(256);
    DeviceState& d = h->d;
    mdx_prof_begin(h, 2);
    switch (mode) {
    case 0: hipLaunchKernelGGL(integrate_kernel<0>, g, b, 0, h->stream, h->S, dt, d.posq, d.vel, d.force, d.ref,
                               d_gate_in, d_disp_out, thr_bits); break;
    case 1: hipLaunchKernelGGL(integrate_kernel<1>, g, b, 0, h->stream, h->S, dt, d.posq, d.vel, d.force, d.ref,
                               d_gate_in, d_disp_out, thr_bits); break;
    default: hipLaunchKernelGGL(integrate_kernel<2>, g, b, 0, h->stream, h->S, dt, d.posq, d.vel, d.force, d.ref,
                                d_gate_in, d_disp_out, thr_bits); break;
    }
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_kinetic(mdx_handle* h) {
    hipLaunchKernelGGL(kinetic_kernel, dim3((h->S + 255) / 256), dim3(256), 0, h->stream, h->S, h->d.vel,
                       h->d.force, h->d.energy, (uint32_t*)(h->d.energy + EN_COUNT));
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}
