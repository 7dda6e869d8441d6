// mdx_integrate.hip — velocity-Verlet kick/drift, the rebuild trigger and the kinetic reduction.
//
// `MdState::step` with `Integrator::VerletVelocity` (/root/reference README.md:237,
// src/ui/panels/md.rs:303-305) as kick(dt/2) - drift(dt) - [forces] - kick(dt/2); inside a
// multi-step burst the closing half kick of step n and the opening half kick of step n+1 are one
// pass (mode 1), so a step streams x, v, f once: R posq+vel+force+ref (64 B), W posq+vel (32 B) + the cleared
// force (16 B) + - dual pair list - ref with the path accumulator (16 B) per slot, one slot per lane, 16-B accesses.
// Pure HBM streaming.
//
// The same pass measures each atom's squared displacement from its position at the last
// neighbour rebuild; a wave whose maximum exceeds (skin/2)^2 raises ctl.disp2[step+1] with an
// atomicMax (non-negative floats order like their bit patterns).  Downstream kernels gate on
// that word.
#include "mdx_internal.h"
#include "mdx_bonded_dev.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

struct LangevinArgs {
    float a1;          // exp(-gamma dt)
    float kt_noise;    // kB T (1 - a1^2): sigma_v^2 = kt_noise * 418.4 / m = kt_noise * vel.w
    uint64_t seed; uint64_t step;   // global step number of this launch
    const uint32_t* orig_of; const uint32_t* gid;
};

// three standard normals for (seed, step, atom): splitmix64 stream keyed by all three, Box-Muller in fp32
// (the oracle draws the same uniforms and evaluates in fp64)
__device__ __forceinline__ void langevin_normals(uint64_t seed, uint64_t step, uint32_t atom, float& g0, float& g1, float& g2) {
    uint64_t st = seed ^ (0x9E3779B97F4A7C15ull * (step + 1ull)) ^ (0xBF58476D1CE4E5B9ull * ((uint64_t)atom + 1ull));
    const float k = 1.0f / 16777216.0f;
    const float u1 = ((float)(mdx_splitmix64(&st) >> 40) + 0.5f) * k, u2 = ((float)(mdx_splitmix64(&st) >> 40) + 0.5f) * k;
    const float u3 = ((float)(mdx_splitmix64(&st) >> 40) + 0.5f) * k, u4 = ((float)(mdx_splitmix64(&st) >> 40) + 0.5f) * k;
    const float r1 = sqrtf(-2.0f * __logf(u1)), r2 = sqrtf(-2.0f * __logf(u3));
    float s1, c1, s2, c2;
    __sincosf(6.2831853f * u2, &s1, &c1); __sincosf(6.2831853f * u4, &s2, &c2);
    g0 = r1 * c1; g1 = r1 * s1; g2 = r2 * c2; (void)s2;
}

// ZERO: clear the force array once it has been consumed - the half-list pair kernel that follows accumulates
// with atomics and needs it zero; doing it here saves a separate fill launch per step.
//
// DUAL (dual pair list): ref[].w accumulates the atom's PATH LENGTH since the last pruning pass of the inner list
// (sum of the step displacements: an upper bound of its displacement, so one scalar per atom suffices and rides
// in the float4 this pass reads anyway; +16 B of writes per slot).  A wave whose maximum exceeds inner_skin/2
// raises the step's prune word; the pair kernel of this step then re-prunes on the spot (mdx_nonbonded.hip).
template <int MODE, bool ZERO, bool DUAL>  // 0: half kick + drift, 1: full kick + drift, 2: closing half kick, 3: Langevin middle
__global__ __launch_bounds__(256) void integrate_kernel(uint32_t S, float dt, float4* __restrict__ posq,
                                                        float4* __restrict__ vel, float4* __restrict__ force,
                                                        float4* __restrict__ ref, const uint32_t* gate_in,
                                                        uint32_t* disp_out, uint32_t thr_bits, LangevinArgs lg,
                                                        uint32_t* prune_out, float path_thr, const uint8_t* __restrict__ skip,
                                                        float* __restrict__ path_arr) {
    const uint32_t gate = gate_in ? *gate_in : 0u;
    if (gate > thr_bits) {  // list already stale: stay a no-op, keep the flag raised
        if (MODE != 2 && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(disp_out, gate);
        return;
    }
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    float d2 = 0.f, path = 0.f;
    // (skip: a solute in rigid water - the waters' slots went through water_step_kernel this step, force cleared and all)
    if (s < S && !(skip && (skip[s] & 4u))) {
        float4 v = vel[s];
        if (v.w != 0.f) {   // w = 418.4/m; 0 marks static, ghost and dummy slots
            const float4 f = force[s];
            const float kdt = ((MODE == 1 || MODE == 3) ? dt : 0.5f * dt) * v.w;
            v.x += kdt * f.x; v.y += kdt * f.y; v.z += kdt * f.z;
            if (MODE != 2) {
                float4 p = posq[s];
                const float ox = p.x, oy = p.y, oz = p.z;
                if (MODE == 3) {
                    const float hdt = 0.5f * dt;
                    p.x += hdt * v.x; p.y += hdt * v.y; p.z += hdt * v.z;
                    float g0, g1, g2;
                    langevin_normals(lg.seed, lg.step, lg.gid[lg.orig_of[s]], g0, g1, g2);
                    const float sig = sqrtf(lg.kt_noise * v.w);
                    v.x = lg.a1 * v.x + sig * g0; v.y = lg.a1 * v.y + sig * g1; v.z = lg.a1 * v.z + sig * g2;
                    p.x += hdt * v.x; p.y += hdt * v.y; p.z += hdt * v.z;
                } else {
                    p.x += dt * v.x; p.y += dt * v.y; p.z += dt * v.z;
                }
                posq[s] = p;
                float4 r = ref[s];
                const float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
                d2 = dx * dx + dy * dy + dz * dz;
                if (!(d2 < 1.0e30f)) d2 = 3.0e38f;  // NaN/inf -> huge, forces a stop
                if (DUAL) {
                    const float mx = p.x - ox, my = p.y - oy, mz = p.z - oz;
                    const float mv = __builtin_sqrtf(mx * mx + my * my + mz * mz);
                    if (path_arr) { path = path_arr[s] + mv; path_arr[s] = path; }      // path split: 8 B instead of the 16-B row written back
                    else { path = r.w + mv; r.w = path; ref[s] = r; }
                }
            }
            vel[s] = v;
        }
        if (ZERO) force[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (MODE != 2) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
        // only a wave that actually crossed the threshold touches the flag word: in the common
        // (not stale) case the pass issues no atomics at all.  One contended atomicMax per wave
        // cost 9x the streaming time of this kernel (~12 ns each, serialised in L2).
        if ((threadIdx.x & 63) == 0 && __float_as_uint(d2) > thr_bits) atomicMax(disp_out, __float_as_uint(d2));
        if (DUAL) {
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) path = fmaxf(path, __shfl_xor(path, m));
            if ((threadIdx.x & 63) == 0 && !(path <= path_thr)) *prune_out = 1u;   // idempotent plain store, rare
        }
    }
}

// Bonded gather + full kick + drift in ONE pass (the large classes' step loop: velocity Verlet inside a chunk, no
// constraints, virtual sites, SPME or external forces, mean role count of a solvent box).  The separate kernels stream
// posq R + roles R + force RMW (bonded gather), then posq RW + vel RW + force R + zero W + ref RW (kick + drift): 226 B per
// water atom in two launches.  Fused, the bonded force of an atom never leaves its registers - v += dt/m (f_pair + f_bonded) -
// and the force array is read once and cleared: ~170 B per atom, one launch.  The gather reads its partners at time t while
// the drift produces t + dt, so positions are double-buffered: read posq_in, write posq_out, the host swaps the two
// pointers (slots that are never integrated - static, ghost, dummy - are copied through).  The bonded terms evaluated here
// are those of the PREVIOUS force call (whose own bonded launch mdx_step left out): positions have not moved since.
// (FusedArgs: mdx_internal.h - the decomposition fills its pipeline fields)
__device__ __forceinline__ void pipe_publish(const FusedArgs& a, uint32_t word) {
    for (uint32_t q = 0; q < a.n_flag; ++q) a.send_buf[a.flag_rows[q]] = make_float4(__uint_as_float(word), 0.f, 0.f, 0.f);
    for (int k = 0; k < 9; ++k) a.pc->m1_shard[k] = 0u;
}
template <bool DUAL, bool PIPE, bool NODIH = false>
__global__ __launch_bounds__(256) void bonded_integrate_kernel(FusedArgs a) {
    uint32_t gate = a.gate_in ? *a.gate_in : 0u;
    if (PIPE && (a.pipe_flags & 1u))       // what the ranks below this one found during the last drift came back with their ghost forces
        for (uint32_t q = 0; q < a.n_flag; ++q) gate = max(gate, __float_as_uint(a.frc_in[a.flag_rows[q]].x));
    if (gate > a.thr_bits) {  // list already stale: stay a no-op.  Positions stay in posq_in; the launcher has swapped the host's pointers
                              // all the same (it enqueues blind), so mdx_step points d.posq back at the buffer the stale step's drift wrote
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicMax(a.disp_out, gate);
            if (PIPE) {
                if (a.gate_word) atomicMax(a.gate_word, gate);
                if (a.pipe_flags & 2u) pipe_publish(a, gate);      // the message still travels, carrying the word; the queues are drawn empty
            }
        }
        return;
    }
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    float d2 = 0.f, path = 0.f;
    if (s < a.S) {
        // three dependent fetches stand between a lane and its bonded force (offsets -> role records -> partner positions):
        // the offsets go first, the lane's own rows travel beside them, and the first FUSED_PRE records with their
        // partners are fetched as one batch each (a loop with a data-dependent trip count would take them one round trip
        // at a time) - a water atom has two or three roles
        const uint32_t rb = a.role_off[s], re = a.role_off[s + 1];
        float4 p = a.posq_in[s];
        float4 v = a.vel[s];
        if (v.w != 0.f) {   // w = 418.4/m; 0 marks static, ghost and dummy slots
            float4 f = a.force[s];
            constexpr int FUSED_PRE = 3;
            RoleRec rr[FUSED_PRE]; float4 q0[FUSED_PRE], q1[FUSED_PRE], prm[FUSED_PRE];
#pragma unroll
            for (int i = 0; i < FUSED_PRE; ++i)      // (past the atom's own list: a null record - slot 0, parameter set 0 - fetched, never evaluated)
                rr[i] = (rb + (uint32_t)i < re) ? a.roles[rb + (uint32_t)i] : RoleRec{{0u, 0u, 0u}, 0u};
#pragma unroll
            for (int i = 0; i < FUSED_PRE; ++i) { q0[i] = a.posq_in[rr[i].p[0]]; q1[i] = a.posq_in[rr[i].p[1]]; prm[i] = a.prm[rr[i].meta >> 8]; }
            RoleEnergies en;
            if (PIPE && (a.pipe_flags & 1u)) {
                const uint32_t nr = a.send_cnt[s];
                for (uint32_t k = 0; k < nr; ++k) { const float4 g = a.frc_in[a.send_rows[(size_t)s * 7 + k]]; f.x += g.x; f.y += g.y; f.z += g.z; }
            }
#pragma unroll
            for (int i = 0; i < FUSED_PRE; ++i)
                if (rb + (uint32_t)i < re) role_compute<false, NODIH>(rr[i], prm[i], p, q0[i], q1[i], a.posq_in, a.p, f.x, f.y, f.z, en);
            for (uint32_t k = rb + FUSED_PRE; k < re; ++k) {
                const RoleRec r = a.roles[k];
                role_eval<false, NODIH>(r, a.prm, p, a.posq_in, a.p, f.x, f.y, f.z, en);
            }
            const float kdt = a.dt * v.w;
            v.x += kdt * f.x; v.y += kdt * f.y; v.z += kdt * f.z;
            const float ox = p.x, oy = p.y, oz = p.z;
            p.x += a.dt * v.x; p.y += a.dt * v.y; p.z += a.dt * v.z;
            if (DUAL && a.path) {
                // path split: |x - ref| <= |x(last pruning pass) - ref| + path since that pass (triangle inequality) - two floats instead
                // of the ref row read and written; the bound is exact at every pruning pass and at most inner_skin / 2 above in between
                const float mx = p.x - ox, my = p.y - oy, mz = p.z - oz;
                path = a.path[s] + __builtin_sqrtf(mx * mx + my * my + mz * mz);
                a.path[s] = path;
                const float d = a.dprune[s] + path;
                d2 = d * d;
                if (!(d2 < 1.0e30f)) d2 = 3.0e38f;  // NaN/inf -> huge, forces a stop
            } else {
                float4 r = a.ref[s];
                const float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
                d2 = dx * dx + dy * dy + dz * dz;
                if (!(d2 < 1.0e30f)) d2 = 3.0e38f;  // NaN/inf -> huge, forces a stop
                if (DUAL) {
                    const float mx = p.x - ox, my = p.y - oy, mz = p.z - oz;
                    path = r.w + __builtin_sqrtf(mx * mx + my * my + mz * mz);
                    r.w = path;
                    a.ref[s] = r;
                }
            }
            a.vel[s] = v;
        }
        a.posq_out[s] = p;
        a.force[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PIPE && (a.pipe_flags & 2u)) {
            const uint32_t nr = a.send_cnt[s];
            for (uint32_t k = 0; k < nr; ++k) a.send_buf[a.send_rows[(size_t)s * 7 + k]] = p;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(d2) > a.thr_bits) atomicMax(a.disp_out, __float_as_uint(d2));
    if (DUAL) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) path = fmaxf(path, __shfl_xor(path, m));
        if ((threadIdx.x & 63) == 0 && !(path <= a.path_thr)) *a.prune_out = 1u;
    }
    if (PIPE && (a.pipe_flags & 2u)) {
        // the workgroup that arrives last knows the stale word is complete: every workgroup's atomicMax (if it issued one) has been
        // acknowledged before it counts itself in.  Two levels (eight shards by blockIdx & 7): ~100 arrivals per word instead of 800.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t shard = blockIdx.x & 7u, in_shard = (gridDim.x - shard + 7u) >> 3;
            if (atomicAdd(a.pc->m1_shard + shard, 1u) + 1u == in_shard) {
                const uint32_t n_shards = gridDim.x < 8u ? gridDim.x : 8u;
                if (atomicAdd(a.pc->m1_shard + 8, 1u) + 1u == n_shards) pipe_publish(a, atomicMax(a.disp_out, 0u));
            }
        }
    }
}

// kinetic energy (kcal/mol) and max |F|^2 over mobile atoms.  Grid-stride over at most 1024 blocks and
// a block-level reduction through LDS: one pair of atomics per block.  (One per wave - 16 k contended
// atomics at 1 M atoms - made this pass take 205 us instead of the 15 us its 33 MB of reads need.)
__global__ __launch_bounds__(256) void kinetic_kernel(uint32_t S, const float4* __restrict__ vel,
                                                      const float4* __restrict__ force, double* energy,
                                                      uint32_t* maxf2_bits) {
    __shared__ double s_ke[4];
    __shared__ float s_f2[4];
    double ke = 0.0;
    float f2 = 0.f;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += gridDim.x * blockDim.x) {
        const float4 v = vel[s];
        if (v.w != 0.f) {
            ke += 0.5 * ((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z) / (double)v.w;
            const float4 f = force[s];
            float q = f.x * f.x + f.y * f.y + f.z * f.z;
            if (!(q < 3.0e38f)) q = 3.0e38f;
            f2 = fmaxf(f2, q);
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        ke += __shfl_xor(ke, m);
        f2 = fmaxf(f2, __shfl_xor(f2, m));
    }
    if ((threadIdx.x & 63) == 0) { s_ke[threadIdx.x >> 6] = ke; s_f2[threadIdx.x >> 6] = f2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ke = s_ke[0] + s_ke[1] + s_ke[2] + s_ke[3];
        f2 = fmaxf(fmaxf(s_f2[0], s_f2[1]), fmaxf(s_f2[2], s_f2[3]));
        if (ke != 0.0) atomicAdd(&energy[EN_KIN], ke);
        if (f2 > 0.f) atomicMax(maxf2_bits, __float_as_uint(f2));
    }
}

// ---- one launch per step (round 6; mdx_nonbonded_impl.h STEP, mdx_api.hip mdx_step "onepass") ---------------------------------------
// A chunk of such steps opens with this pass: positions go into the step form for a step that opens with a HALF kick - Y = x + dt v,
// the force rows the first launch reads keep the complete forces and get .w = dt^2 418.4/m / 2, so that Y + w F is x + dt (v + dt/2 a) -
// the two other force buffers are zeroed with the full .w, and the words that gate the first launch are raised from that very position:
// nothing is predicted here.
__global__ __launch_bounds__(256) void step_begin_kernel(uint32_t S, float dt, const float4* __restrict__ x_in, float4* __restrict__ y_out,
                                                         const float4* __restrict__ vel, float4* __restrict__ f0, float4* __restrict__ fb,
                                                         float4* __restrict__ fc, const float* __restrict__ path, const float4* __restrict__ ref,
                                                         const uint32_t* __restrict__ gate_in,
                                                         uint32_t* __restrict__ disp_out, uint32_t thr_bits, uint32_t* __restrict__ prune_out,
                                                         float path_thr) {
    const uint32_t gate = gate_in ? *gate_in : 0u;
    if (gate > thr_bits) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(disp_out, gate);
        return;
    }
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    float d2 = 0.f, pw = 0.f;
    if (s < S) {
        float4 p = x_in[s];
        const float4 v = vel[s];
        float w = 0.f;
        if (v.w != 0.f) {
            float4 f = f0[s];
            w = dt * dt * v.w;
            f.w = 0.5f * w;
            const float kx = f.w * f.x, ky = f.w * f.y, kz = f.w * f.z;
            const float mx = fmaf(dt, v.x, kx), my = fmaf(dt, v.y, ky), mz = fmaf(dt, v.z, kz);      // the step's displacement, as the first launch will find it
            const float4 r = ref[s];
            const float ex = p.x + mx - r.x, ey = p.y + my - r.y, ez = p.z + mz - r.z;      // the stage the first launch will reconstruct, against the list's reference
            p.x = fmaf(dt, v.x, p.x); p.y = fmaf(dt, v.y, p.y); p.z = fmaf(dt, v.z, p.z);
            pw = path[s] + __builtin_sqrtf(mx * mx + my * my + mz * mz);
            d2 = ex * ex + ey * ey + ez * ez;
            if (!(d2 < 1.0e30f)) d2 = 3.0e38f;
            f0[s] = f;
        } else f0[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        y_out[s] = p;
        const float4 z = make_float4(0.f, 0.f, 0.f, w);
        fb[s] = z; fc[s] = z;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { d2 = fmaxf(d2, __shfl_xor(d2, m)); pw = fmaxf(pw, __shfl_xor(pw, m)); }
    if ((threadIdx.x & 63) == 0) {
        if (__float_as_uint(d2) > thr_bits) atomicMax(disp_out, __float_as_uint(d2));
        if (prune_out && !(pw <= path_thr)) *prune_out = 1u;
    }
}

// ... and when a launch of the chunk found itself gated off (the list went stale at its position stage) or a kick exceeded what had been
// granted, the host takes the state back into the plain form: x = Y + w F, v += dt 418.4/m F unless the launch that raised the flag had
// already kicked (kick_done).
__global__ __launch_bounds__(256) void step_materialise_kernel(uint32_t S, float dt, const float4* __restrict__ y, const float4* __restrict__ fprev,
                                                               float4* __restrict__ vel, float4* __restrict__ x_out, float* __restrict__ path,
                                                               uint32_t kick_done, float kick, const uint32_t* __restrict__ gate_in, uint32_t thr_bits) {
    if (gate_in && *gate_in > thr_bits) return;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float4 Y = y[s], F = fprev[s];
    float4 v = vel[s];
    if (v.w != 0.f) {
        x_out[s] = step_pos(Y, F);
        if (!kick_done) {      // (else the launch that kicked also counted the path)
            const float kdt = kick * dt * v.w;      // (kick: 0.5 for the chunk's first step, whose force rows carry w / 2)
            const float sx = fmaf(dt, v.x, F.w * F.x), sy = fmaf(dt, v.y, F.w * F.y), sz = fmaf(dt, v.z, F.w * F.z);      // this drift, for the path length
            path[s] += __builtin_sqrtf(sx * sx + sy * sy + sz * sz);
            v.x = fmaf(kdt, F.x, v.x); v.y = fmaf(kdt, F.y, v.y); v.z = fmaf(kdt, F.z, v.z);
            vel[s] = v;
        }
    } else x_out[s] = Y;
}

int mdx_launch_step_begin(mdx_handle* h, float dt, const uint32_t* d_gate_in, uint32_t* d_disp_out, uint32_t thr_bits, uint32_t* d_prune_out) {
    DeviceState& d = h->d;
    const dim3 g((h->S + 255) / 256), b(256);
    mdx_prof_begin(h, 2);
    hipLaunchKernelGGL(step_begin_kernel, g, b, 0, h->stream, h->S, dt, d.posq, d.posq_alt, d.vel, d.force, d.force_b, d.force_c, d.path, d.ref,
                       d_gate_in, d_disp_out, thr_bits, d_prune_out, 0.5f * h->inner_skin * (1.0f - 1.0e-4f));
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_step_materialise(mdx_handle* h, float dt, const float4* y, const float4* fprev, float4* x_out, bool kick_done, float kick,
                                const uint32_t* d_gate_in, uint32_t thr_bits) {
    const dim3 g((h->S + 255) / 256), b(256);
    hipLaunchKernelGGL(step_materialise_kernel, g, b, 0, h->stream, h->S, dt, y, fprev, h->d.vel, x_out, h->d.path, kick_done ? 1u : 0u, kick, d_gate_in, thr_bits);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_integrate(mdx_handle* h, int mode, float dt, const uint32_t* d_gate_in, uint32_t* d_disp_out,
                         uint32_t thr_bits, uint32_t* d_prune_out, bool skip_wstep) {
    const uint8_t* const skip = (skip_wstep && mode != 2) ? h->d.wstep_s : nullptr;
    float* const path_arr = h->path_split ? h->d.path : nullptr;
    const dim3 g((h->S + 255) / 256), b(256);
    DeviceState& d = h->d;
    if (mode != 2) h->vsites_fresh = false;      // a drift: the virtual sites follow at the next position stage / construct launch
    mdx_prof_begin(h, 2);
    LangevinArgs lg{};
    if (mode == 3) {
        lg.a1 = std::exp(-h->lang_gamma * dt);
        lg.kt_noise = (float)(MDX_KB * (double)h->lang_temp * (1.0 - (double)lg.a1 * lg.a1));
        lg.seed = h->lang_seed; lg.step = h->lang_step;
        lg.orig_of = d.orig_of; lg.gid = d.gid;
    }
    // the force array is consumed here and rebuilt by the force pass that follows modes 0/1/3
    const bool zero = mode != 2 && mdx_nb_half(h);
    // dual list: only the drift passes of the step loop accumulate path lengths (d_prune_out set by mdx_step)
    const bool dual = h->dual_on && d_prune_out != nullptr && mode != 2 && zero;
    const float path_thr = 0.5f * h->inner_skin * (1.0f - 1.0e-4f);
#define INTEG(M)                                                                                                      \
    do {                                                                                                              \
        if (dual) hipLaunchKernelGGL((integrate_kernel<M, true, true>), g, b, 0, h->stream, h->S, dt, d.posq, d.vel, d.force, \
                                     d.ref, d_gate_in, d_disp_out, thr_bits, lg, d_prune_out, path_thr, skip, path_arr); \
        else if (zero) hipLaunchKernelGGL((integrate_kernel<M, true, false>), g, b, 0, h->stream, h->S, dt, d.posq, d.vel, d.force, \
                                     d.ref, d_gate_in, d_disp_out, thr_bits, lg, nullptr, 0.f, skip, nullptr);        \
        else hipLaunchKernelGGL((integrate_kernel<M, false, false>), g, b, 0, h->stream, h->S, dt, d.posq, d.vel, d.force,  \
                                d.ref, d_gate_in, d_disp_out, thr_bits, lg, nullptr, 0.f, skip, nullptr);             \
    } while (0)
    switch (mode) {
    case 0: INTEG(0); break;
    case 1: INTEG(1); break;
    case 3: INTEG(3); break;
    default: INTEG(2); break;
    }
    if (zero) h->force_zeroed = true;
#undef INTEG
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

// MDX_FUSE_BONDED_INTEGRATE=0: A/B knob (the separate bonded gather and kick + drift launches at every step)
bool mdx_bonded_integrate_ok(const mdx_handle* h) {
    const char* e = std::getenv("MDX_FUSE_BONDED_INTEGRATE");     // read per chunk, so one process can compare both arrangements
    const bool on = !(e && e[0] == '0');
    // decomposed handles (round 3): ghost slots carry w = 0 and are copied through, the halo unpack writes into the buffer the
    // pass has just filled; MDX_FUSE_BONDED_INTEGRATE_DD=0 keeps the separate launches there
    static const bool dd_env = [] { const char* x = std::getenv("MDX_FUSE_BONDED_INTEGRATE_DD"); return !(x && x[0] == '0'); }();
    const bool dd_ok = dd_env && h->dd != nullptr;
    // (n_roles counts the whole system; a decomposed handle holds n_local of its N atoms)
    const double roles_here = (double)h->n_roles * ((h->n_local != h->N && h->N) ? (double)h->n_local / (double)h->N : 1.0);
    return on && mdx_nb_variant(h) >= 2 && h->integrator == MDX_INTEGRATOR_VERLET_VELOCITY && !mdx_has_constraints(h) && h->n_vsites == 0 &&
           !h->pme_on && !h->have_ext && (dd_ok || (!h->dd && h->n_local == h->N)) && !h->alch_on && mdx_bonded_wanted(h) && h->T >= mdx_wpt8_below(h) &&
           roles_here < 2.6 * (double)h->S && h->d.posq_alt != nullptr;
}

int mdx_launch_bonded_integrate(mdx_handle* h, float dt, const uint32_t* d_gate_in, uint32_t* d_disp_out, uint32_t thr_bits,
                                uint32_t* d_prune_out) {
    DeviceState& d = h->d;
    FusedArgs a{};
    a.S = h->S; a.dt = dt; a.posq_in = d.posq; a.posq_out = d.posq_alt; a.vel = d.vel; a.force = d.force; a.ref = d.ref;
    a.path = h->path_split ? d.path : nullptr; a.dprune = h->path_split ? d.dprune : nullptr;
    a.role_off = d.role_off_s; a.roles = d.role_rec_s; a.prm = d.role_prm; a.R = h->n_roles;
    mdx_fill_bonded_params(h, a.p);
    a.gate_in = d_gate_in; a.disp_out = d_disp_out; a.thr_bits = thr_bits; a.prune_out = d_prune_out;
    a.path_thr = 0.5f * h->inner_skin * (1.0f - 1.0e-4f);
    const bool dual = h->dual_on && d_prune_out != nullptr;
    const dim3 g((h->S + 255) / 256), b(256);
    const bool pipe = mdx_dd_pipe_fill(h, a, const_cast<uint32_t*>(d_gate_in));      // decomposed handle, pipelined arrangement: returned ghost forces in, halo pack out
    mdx_prof_begin(h, 5);
    // no dihedral role in the system (a box of flexible water): the flavour without that branch - 67 VGPRs instead of 93, seven waves per
    // SIMD instead of five for a pass that lives on loads in flight.  MDX_FUSED_NODIH=0: A/B (read per launch)
    const char* nd_env = std::getenv("MDX_FUSED_NODIH");
    const bool nodih = h->n_roles_dih == 0 && !(nd_env && nd_env[0] == '0');
    if (pipe) {
        if (dual) { if (nodih) hipLaunchKernelGGL((bonded_integrate_kernel<true, true, true>), g, b, 0, h->stream, a); else hipLaunchKernelGGL((bonded_integrate_kernel<true, true>), g, b, 0, h->stream, a); }
        else hipLaunchKernelGGL((bonded_integrate_kernel<false, true>), g, b, 0, h->stream, a);
    } else {
        if (dual) { if (nodih) hipLaunchKernelGGL((bonded_integrate_kernel<true, false, true>), g, b, 0, h->stream, a); else hipLaunchKernelGGL((bonded_integrate_kernel<true, false>), g, b, 0, h->stream, a); }
        else hipLaunchKernelGGL((bonded_integrate_kernel<false, false>), g, b, 0, h->stream, a);
    }
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    std::swap(d.posq, d.posq_alt);
    h->force_zeroed = true;
    return MDX_OK;
}

int mdx_launch_kinetic(mdx_handle* h) {
    hipLaunchKernelGGL(kinetic_kernel, dim3(std::min<uint32_t>((h->S + 255) / 256, 1024u)), dim3(256), 0, h->stream, h->S, h->d.vel,
                       h->d.force, h->d.energy, (uint32_t*)(h->d.energy + EN_COUNT));
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}
