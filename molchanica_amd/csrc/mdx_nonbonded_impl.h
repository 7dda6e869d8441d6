// mdx_nonbonded_impl.h - the pair kernels (templates) and their launcher launch_variant<ENERGY, COUL>; the description is at the head of
// mdx_nonbonded.hip.  Included by mdx_nb_inst.hip, which the Makefile compiles once per (ENERGY, COUL) pair: the 650-odd kernel
// instantiations build as eight translation units side by side (one unit took 3.5 minutes).
#pragma once
#include "mdx_internal.h"
#include "mdx_bonded_dev.h"
#include "mdx_pair_dev.h"
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <type_traits>

constexpr int NB_WAVES = MDX_NB_WAVES;      // least waves per workgroup (mdx_internal.h)
#ifndef NB_ASM_PAIR
#define NB_ASM_PAIR 0     // 1: the default flavour's pair evaluation is the hand-written block pair_eval_asm (v_cmpx exec handling):
                          // measured round 3 - inner-list walk 0.4566 / 0.4517 ms against 0.4565 / 0.4572 with the compiler's
                          // v_cmp + s_and_saveexec + s_cbranch_execz + s_or, pruning pass SLOWER (no early-out for the 40 % of its
                          // cluster pairs without a lane in range): 0.4936 vs 0.4809 ms per step.  Two scalar instructions fewer per
                          // cluster pair buy nothing; kept as an A/B arm (make EXTRA=-DNB_ASM_PAIR=1), parity-tested once
#endif
#ifndef NB_HALF_FLUSH
#define NB_HALF_FLUSH 0   // 1: j-forces leave once per chunk from LDS; 0: one 24-lane atomic per entry
#endif

// The force-only pair evaluation of the default flavour (shifted-cutoff Coulomb, Lorentz-Berthelot, one cutoff, half list) as ONE
// block of hand-written gfx950 instructions (round 3).  Same arithmetic, instruction for instruction, as hipcc's code for
// pair_eval<..., BRANCHY, HALF>; what differs is the exec-mask handling.  The compiler brackets the in-range part with
//     v_cmp_gt_f32 vcc / s_and_saveexec_b64 / s_cbranch_execz ... s_or_b64 exec
// and this loop is issue-bound with scalar instructions costing as much as vector ones (65 M SALU are
// 17 % of the launch).  Here v_cmpx_gt_f32 narrows EXEC itself and one s_mov_b64 restores it: one scalar instruction per cluster
// pair instead of three.  Preconditions, all true in nb_cluster_body's entry loop: EXEC is all ones on entry (whole waves, no
// divergent region around the call); nothing after the block reads VCC.  Hazards: the only EXEC write by a VALU instruction is
// the v_cmpx, and 21 VALU instructions separate it from whatever follows the block (DPP and v_readlane want 5 / 4 wait states
// after a VALU write of EXEC; an s_mov of EXEC needs none); the first use of the v_rsq result is four instructions behind it.  BIAS: the masked chunks' NaN-coded exclusion
// bit enters r^2 as the addend of the first FMA.  R2OUT: the pruning pass wants r^2 (of every lane, in range or not).
template <bool BIAS, bool R2OUT>
__device__ __forceinline__ void pair_eval_asm(float xi, float yi, float zi, float qi, float sgi, float epi, const float4 pj,
                                              const float2 lj, float rc2, float bias, float& fx, float& fy, float& fz,
                                              float& g0, float& g1, float& g2, float& r2_out) {
    float dx, dy, dz, r2, t1, t2, t3, t4;
    if (BIAS) {
        asm volatile(
            "v_sub_f32 %[dy], %[yi], %[yj]\n\t"
            "v_sub_f32 %[dx], %[xi], %[xj]\n\t"
            "v_fma_f32 %[r2], %[dy], %[dy], %[bias]\n\t"
            "v_sub_f32 %[dz], %[zi], %[zj]\n\t"
            "v_fmac_f32 %[r2], %[dx], %[dx]\n\t"
            "v_fmac_f32 %[r2], %[dz], %[dz]\n\t"
            : [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [r2] "=&v"(r2)
            : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [xj] "v"(pj.x), [yj] "v"(pj.y), [zj] "v"(pj.z), [bias] "v"(bias));
    } else {
        asm volatile(
            "v_sub_f32 %[dy], %[yi], %[yj]\n\t"
            "v_sub_f32 %[dx], %[xi], %[xj]\n\t"
            "v_mul_f32 %[r2], %[dy], %[dy]\n\t"
            "v_sub_f32 %[dz], %[zi], %[zj]\n\t"
            "v_fmac_f32 %[r2], %[dx], %[dx]\n\t"
            "v_fmac_f32 %[r2], %[dz], %[dz]\n\t"
            : [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [r2] "=&v"(r2)
            : [xi] "v"(xi), [yi] "v"(yi), [zi] "v"(zi), [xj] "v"(pj.x), [yj] "v"(pj.y), [zj] "v"(pj.z));
    }
    if (R2OUT) r2_out = r2;
    asm volatile(
        "v_cmpx_gt_f32 vcc, %[rc2], %[r2]\n\t"            // EXEC &= (r^2 < rc^2); a NaN r^2 (excluded pair) fails
        "v_rsq_f32 %[r2], %[r2]\n\t"                      // 1 / r
        "v_add_f32 %[t1], %[sgi], %[ljx]\n\t"             // sigma_ij = sigma_i / 2 + sigma_j / 2
        "v_mul_f32 %[t1], %[t1], %[t1]\n\t"
        "v_mul_f32 %[t2], %[epi], %[ljy]\n\t"             // 24 eps_ij
        "v_mul_f32 %[t3], %[r2], %[r2]\n\t"               // 1 / r^2
        "v_mul_f32 %[t1], %[t1], %[t3]\n\t"               // s2 = sigma^2 / r^2
        "v_mul_f32 %[t4], %[t1], %[t1]\n\t"
        "v_mul_f32 %[t1], %[t1], %[t4]\n\t"               // s6
        "v_mul_f32 %[t2], %[t2], %[t1]\n\t"               // 24 eps s6
        "v_fma_f32 %[t1], %[t1], 2.0, -1.0\n\t"           // 2 s6 - 1
        "v_mul_f32 %[t1], %[t2], %[t1]\n\t"               // LJ force * r^2
        "v_mul_f32 %[t2], %[qi], %[qj]\n\t"               // k_e q_i q_j
        "v_fmac_f32 %[t1], %[t2], %[r2]\n\t"              // + Coulomb force * r^2
        "v_mul_f32 %[t1], %[t3], %[t1]\n\t"               // fs = (...) / r^2
        "v_fmac_f32 %[fx], %[dx], %[t1]\n\t"
        "v_fmac_f32 %[fy], %[dy], %[t1]\n\t"
        "v_fmac_f32 %[fz], %[dz], %[t1]\n\t"
        "v_fmac_f32 %[g0], %[dx], %[t1]\n\t"
        "v_fmac_f32 %[g1], %[dy], %[t1]\n\t"
        "v_fmac_f32 %[g2], %[dz], %[t1]\n\t"
        "s_mov_b64 exec, -1\n\t"
        : [fx] "+v"(fx), [fy] "+v"(fy), [fz] "+v"(fz), [g0] "+v"(g0), [g1] "+v"(g1), [g2] "+v"(g2), [r2] "+v"(r2),
          [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4)
        : [rc2] "s"(rc2), [sgi] "v"(sgi), [epi] "v"(epi), [ljx] "v"(lj.x), [ljy] "v"(lj.y), [qi] "v"(qi), [qj] "v"(pj.w),
          [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz)
        : "vcc");
}

template <bool ENERGY, int COUL, bool GEOM, bool SAMECUT, bool MASKED>
__device__ __forceinline__ void chunk_pairs(const float4* __restrict__ sx, const float2* __restrict__ sl,
                                            unsigned long long mask, float xi, float yi, float zi, float qi,
                                            float sgi, float epi, const NbParams& p, float& fx, float& fy,
                                            float& fz, double& elj, double& ecoul, double& evir) {
    float e1 = 0.f, e2 = 0.f, e3 = 0.f;
#pragma unroll 8
    for (int jj = 0; jj < 64; ++jj) {
        const bool allowed = MASKED ? (bool)((mask >> jj) & 1ull) : true;
        pair_eval<ENERGY, COUL, GEOM, SAMECUT, false>(xi, yi, zi, qi, sgi, epi, sx[jj], sl[jj], allowed, p, fx, fy, fz,
                                               e1, e2, nullptr, ENERGY ? &e3 : nullptr);
    }
    if (ENERGY) { elj += (double)e1; ecoul += (double)e2; evir += (double)e3; }
}

template <bool ENERGY, int COUL, bool GEOM, bool SAMECUT>
__global__ __launch_bounds__(NB_WAVES * 64) void nb_tile_kernel(NbArgs a) {
    if (a.gate && *a.gate > a.thr_bits) return;
    __shared__ float4 s_xyzq[NB_WAVES][64];
    __shared__ float2 s_lj[NB_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware remap: block b is dispatched to XCD b%8; give each XCD a contiguous tile range
    const uint32_t nblocks = (a.T + NB_WAVES - 1) / NB_WAVES;
    const uint32_t per_xcd = (nblocks + 7) >> 3;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= nblocks) return;
    const uint32_t t = blk * NB_WAVES + wave;
    if (t >= a.T) return;

    const uint32_t islot = t * MDX_TILE + lane;
    const float4 pi = a.posq[islot];
    const float2 li = a.lj[islot];
    const ListCounts cnt = a.counts[t];
    const uint32_t e0 = a.entry_off[t];
    const uint32_t nmc = cnt.n_masked >> 3, nchunks = (cnt.n_masked + cnt.n_plain) >> 3;
    const uint32_t mbase = a.mchunk_off[t];
    float4* sx = s_xyzq[wave];
    float2* sl = s_lj[wave];

    float fx = 0.f, fy = 0.f, fz = 0.f;
    double elj = 0.0, ecoul = 0.0, evir = 0.0;

    // prefetch chunk 0
    float4 nj = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 nl = make_float2(0.f, 0.f);
    uint32_t ncode = 13;
    if (nchunks) {
        const uint2 ent = a.entries[e0 + (lane >> 3)];
        const uint32_t js = ent.x * MDX_CLUSTER + (lane & 7);
        nj = a.posq[js]; nl = a.lj[js]; ncode = ent.y & 31u;
    }
    for (uint32_t c = 0; c < nchunks; ++c) {
        {   // image shift, then park in LDS
            const int kx = (int)(ncode % 3u) - 1, ky = (int)((ncode / 3u) % 3u) - 1, kz = (int)(ncode / 9u) - 1;
            nj.x += (float)kx * a.p.shift[0];
            nj.y += (float)ky * a.p.shift[1];
            nj.z += (float)kz * a.p.shift[2];
            sx[lane] = nj;
            sl[lane] = nl;
        }
        WAVE_LDS_SYNC();
        if (c + 1 < nchunks) {
            const uint2 ent = a.entries[e0 + (c + 1) * 8 + (lane >> 3)];
            const uint32_t js = ent.x * MDX_CLUSTER + (lane & 7);
            nj = a.posq[js]; nl = a.lj[js]; ncode = ent.y & 31u;
        }
        if (c < nmc) {
            const unsigned long long m = a.masks[(size_t)(mbase + c) * 64 + lane];
            chunk_pairs<ENERGY, COUL, GEOM, SAMECUT, true>(sx, sl, m, pi.x, pi.y, pi.z, pi.w, li.x, li.y, a.p,
                                                           fx, fy, fz, elj, ecoul, evir);
        } else {
            chunk_pairs<ENERGY, COUL, GEOM, SAMECUT, false>(sx, sl, ~0ull, pi.x, pi.y, pi.z, pi.w, li.x, li.y,
                                                            a.p, fx, fy, fz, elj, ecoul, evir);
        }
        WAVE_LDS_SYNC();
    }
    a.force[islot] = make_float4(fx, fy, fz, 0.f);
    if (ENERGY) {
        if (!(a.slot_flags[islot] & 2u)) { elj = 0.0; ecoul = 0.0; evir = 0.0; }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            elj += __shfl_xor(elj, m);
            ecoul += __shfl_xor(ecoul, m);
            evir += __shfl_xor(evir, m);
        }
        if (lane == 0) {   // every pair is seen from both sides
            double* q = a.energy + EN_COUNT + 8 + MDX_ESTRIDE * (blk & (MDX_EPART - 1));
            atomicAdd(q, 0.5 * elj); atomicAdd(q + 1, 0.5 * ecoul); atomicAdd(q + 2, 0.5 * evir);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Variant 2 (default): cluster-masked tile kernel.  Same list, same LDS staging, but a lane is the
// pair (i-atom ii = lane&7 of every i-cluster, j-atom jj = lane>>3 of the current entry): the
// eight i-clusters of the tile sit in registers, and an entry is evaluated only against the
// i-clusters whose bounding box is within the list radius of the j-cluster (the entry's 8-bit
// imask, a wave-uniform branch).  That removes the corner (i-cluster, j-cluster) pairs a
// whole-tile test lets through - about 40 % of the pair evaluations at rc 10 + skin 2 - at the
// price of one cross-lane reduction per tile.  Still one owner per i-atom, no atomics,
// deterministic.
//
// WPT = waves per tile.  WPT = 1: a workgroup is 4 tiles.  WPT = 4: the 4 waves of a workgroup
// share ONE tile, wave w takes chunks w, w+4, ... of its list and the partial forces are summed
// through LDS in a fixed order.  A tile's list is a ~270 us dependency chain for one wave, so a
// launch with fewer tiles than the chip has wave slots (strong scaling: 1/8 of the box per GPU,
// or any system below ~250 k atoms) is latency-bound; splitting the list 4 ways fills the SIMDs.
//
// HALF = true (variant 5) walks a half list: a cluster pair lives in ONE tile's list, the force on
// the i-atoms accumulates in registers as before and the reaction on the eight j-atoms of an entry
// is summed over the eight i-lanes with three DPP steps and leaves as ONE 24-lane f32 atomic
// (x, y, z of 8 consecutive float4 records = 128 contiguous bytes).  Measured on gfx950
// (tools/ubench/atomic_jforce.hip, atomic_flush.hip): the L2 retires ~250 G f32 atomic words/s, so the
// 88 M words of a 1 M-atom half list need 0.35 ms of it - hidden behind 0.6 ms of arithmetic as long
// as no instruction shape hits the same 128-B line twice in a row (x, then y, then z of the same
// atoms is 3x slower).  Measured and rejected: collecting a chunk's j-forces in LDS and flushing
// them once per chunk (NB_HALF_FLUSH=1: 0.79 vs 0.66 ms, it needs 137 VGPRs = 3 waves/SIMD);
// any form of LDS look-ahead of the next entry's j record (+6..8 %, both kernels).
// The force array must be zero when the kernel starts.
template <int CTRL>
__device__ __forceinline__ float dpp_xadd(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// DUAL (dual pair list, half-list force flavour only): 0 = plain list; 1 = walk the INNER list, a no-op when this step
// must prune; 2 = the pruning pass (walks the Verlet list, writes the inner list), a no-op unless this step must prune.  The step loop enqueues 1 and 2 back to back
// and the device decides which one runs (a gated-off launch costs ~3 us).  A single kernel with a run-time switch was
// measured first: the two scalar instructions it adds per cluster pair cost 11 % (0.537 -> 0.598 ms) - every
// instruction in this loop is ~5 cycles of a latency-bound wave.
// The body of the cluster kernel.  DUAL here is 0, 1 or 2; the __global__ wrapper below owns the LDS arrays and, for the
// merged dual-list launch (DUAL 3), picks the inner-walk or the pruning body at run time: one launch per step instead
// of a pair of which the device runs one (the gated-off twin cost ~4 us per step - 5 % of a 23 k-atom step).
// STEP (round 6, "one launch per step": NbArgs::st_*): the tile's wave first FINISHES THE LAST STEP for its 64 atoms - position x = Y + w F
// from the two rows every reader uses (mdx_bonded_dev.h step_pos), full kick, the next step's Y, path length, the words that gate the next
// launch - then evaluates the pairs at x as below, j-atoms reconstructed from the same two rows, and its atoms' bonded roles.  The separate
// bonded + kick + drift pass (35 us of a 513 us step at 1 M atoms, 112 B per atom) rides inside this compute-bound launch instead.
template <bool ENERGY, int COUL, bool GEOM, bool SAMECUT, int WPT, bool HALF, bool ALCH, int DUAL, int BW, bool STEP = false>
__device__ __forceinline__ void nb_cluster_body(const NbArgs& a, const bool owned_prune, float4 (*s_xyzq)[64], float2 (*s_lj)[64],
                                                float (*s_red)[3][64], float (*s_ownj)[64], float4 (*s_g)[64],
                                                unsigned long long (*s_mask)[64], const float4* __restrict__ etab = nullptr) {
    // the wave index is wave-uniform: say so, and tile number, list bounds and the chunk loop
    // live in SGPRs with scalar branches instead of VGPR compares and exec-mask loops
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // BW < WPT: the WPT waves of a tile sit in WPT / BW workgroups (systems of a few hundred tiles: with one workgroup per
    // tile 370 tiles land on 256 CUs as one or two per CU and the launch lasts as long as the CUs that got two)
    constexpr int SPLIT = BW < WPT ? WPT / BW : 1;
    static_assert(SPLIT == 1 || HALF, "a tile split over workgroups returns its i-forces through atomics");
    constexpr int TPB = SPLIT > 1 ? 1 : BW / WPT;             // tiles per workgroup
    const uint32_t ntiles = a.tile_order ? a.t_count : a.T;   // (a decomposed handle launches its interior and boundary tiles apart)
    const uint32_t nblocks = SPLIT > 1 ? ntiles * SPLIT : (ntiles + TPB - 1) / TPB;
    const uint32_t per_xcd = (nblocks + 7) >> 3;
    // a decomposed rank's tile range is owned bricks (long lists) and halo shells (short lists) in spatial order: a
    // contiguous eighth per XCD would leave whole XCDs with halo tiles only, so there the tiles go round-robin
    const uint32_t blk = a.xcd_interleave ? blockIdx.x : (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= nblocks) return;                               // whole workgroup
    const int tib = SPLIT > 1 ? 0 : wave / WPT;               // tile within the workgroup
    const int part = SPLIT > 1 ? (int)(blk % SPLIT) * BW + wave : wave % WPT;   // which share of the tile's chunks
    const uint32_t tidx = SPLIT > 1 ? blk / SPLIT : blk * TPB + tib;
    const bool t_ok = tidx < ntiles;                          // (the last workgroup may have a tile too many)
    if (WPT == 1 && !t_ok) return;
    // a.T = the null tile: empty list, nothing stored
    const uint32_t t = t_ok ? (a.tile_order ? __builtin_amdgcn_readfirstlane(a.tile_order[a.t_first + tidx]) : tidx) : a.T;
    const int ii = lane & 7, jj = lane >> 3;

    float xi[8], yi[8], zi[8], qi[8], sgi[8], epi[8], fx[8], fy[8], fz[8];
    if constexpr (STEP) {
        static_assert(HALF && !ENERGY && !ALCH && DUAL != 0, "one launch per step: half list, dual list");
        // lane = slot.  Every wave of the tile reconstructs the tile's positions (it needs them as i-atoms); the tile's FIRST wave also
        // finishes the last step for them: kick, next Y, path length, the words of the next launch.
        const uint32_t s = t * MDX_TILE + lane;
        const float4 Y = a.posq[s], Fp = a.st_fprev[s];
        const float4 x = step_pos(Y, Fp);        // (static and dummy slots carry w = 0)
        if (part == 0 && t_ok) {
            float4 v = a.st_vel[s];
            float d2w = 0.f, pw = 0.f;       // what this lane contributes to the words of the NEXT launch
            bool viol = false;
            if (v.w != 0.f) {   // w = 418.4/m; 0 marks static and dummy slots
                // (the chunk's first launch finishes a step that opens with a HALF kick: its force rows carry w / 2, st_kick is 0.5)
                const float w = Fp.w, kdt = a.st_kick * a.st_dt * v.w;
                const float4 Fm = a.st_fnext[s];                                // the forces of the stage before Fp's (the buffer this launch zeroes)
                const float kx = w * Fp.x, ky = w * Fp.y, kz = w * Fp.z;       // the kick's share of this step's displacement
                const float sx_ = fmaf(a.st_dt, v.x, kx), sy_ = fmaf(a.st_dt, v.y, ky), sz_ = fmaf(a.st_dt, v.z, kz);      // x - x(previous stage)
                v.x = fmaf(kdt, Fp.x, v.x); v.y = fmaf(kdt, Fp.y, v.y); v.z = fmaf(kdt, Fp.z, v.z);
                float path;
                const float4 r = a.ref[s];
                const float dpr = __builtin_sqrtf((x.x - r.x) * (x.x - r.x) + (x.y - r.y) * (x.y - r.y) + (x.z - r.z) * (x.z - r.z));      // |x - ref|, exactly
                if (DUAL == 2) {   // this launch is a pruning pass: the inner list it writes is exact at x, paths count from here
                    path = 0.f;
                    a.dprune[s] = dpr;
                } else path = a.path[s] + __builtin_sqrtf(sx_ * sx_ + sy_ * sy_ + sz_ * sz_);
                a.path[s] = path;
                // The words that let this launch run were raised by the launch before it from Y and a GRANT for the kick it could not know
                // (below).  Now the stage is known exactly: had the words been raised from it, would this launch walk the inner list although
                // the path budget is spent, or run at all although the list is stale?  Then its forces are not to be trusted - the host
                // takes the step back to a list rebuild (rare: the atom that decides the word must also be the one that outran its grant).
                {
                    float dt2 = dpr * dpr;
                    if (!(dt2 < 1.0e30f)) dt2 = 3.0e38f;
                    viol = (DUAL != 2 && !(path <= a.st_path_thr)) || __float_as_uint(dt2) > a.thr_bits;
                }
                float4 yo = x;
                if (!a.st_last) {
                    yo.x = fmaf(a.st_dt, v.x, x.x); yo.y = fmaf(a.st_dt, v.y, x.y); yo.z = fmaf(a.st_dt, v.z, x.z);
                    // The next stage is yo + w F(x), F not known before this launch ends: the words that gate the next launch grant the
                    // kick w (|2 Fp - Fm| + |Fp - Fm| + 30 kcal/mol/A) - the force extrapolated from the last two stages and room for its
                    // curvature; ~0.01 A for a water hydrogen at 0.5 fs - and the next launch checks the words against what the stage turned out to be.
                    const float wf = a.st_dt * a.st_dt * v.w;
                    const float ex = 2.f * Fp.x - Fm.x, ey = 2.f * Fp.y - Fm.y, ez = 2.f * Fp.z - Fm.z;
                    const float gx = Fp.x - Fm.x, gy = Fp.y - Fm.y, gz = Fp.z - Fm.z;
                    const float m = a.st_grant * wf * (__builtin_sqrtf(ex * ex + ey * ey + ez * ez) + __builtin_sqrtf(gx * gx + gy * gy + gz * gz) + 30.f);
                    pw = path + a.st_dt * __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z) + m;
                    const float db = __builtin_sqrtf((yo.x - r.x) * (yo.x - r.x) + (yo.y - r.y) * (yo.y - r.y) + (yo.z - r.z) * (yo.z - r.z)) + m;
                    d2w = db * db;
                    if (!(d2w < 1.0e30f)) d2w = 3.0e38f;      // NaN / inf -> huge, forces a stop
                }
                a.st_vel[s] = v;
                a.st_yout[s] = yo;
            } else a.st_yout[s] = Y;
            a.st_fnext[s] = make_float4(0.f, 0.f, 0.f, a.st_dt * a.st_dt * v.w);
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) { d2w = fmaxf(d2w, __shfl_xor(d2w, m)); pw = fmaxf(pw, __shfl_xor(pw, m)); }
            const bool any_viol = __any(viol);
            if (lane == 0) {
                if (a.st_disp_out && __float_as_uint(d2w) > a.thr_bits) atomicMax(a.st_disp_out, __float_as_uint(d2w));
                if (a.st_prune_out && !(pw <= a.st_path_thr)) *a.st_prune_out = 1u;
                if (any_viol) { *a.st_viol = 1u; if (a.st_disp_out) atomicMax(a.st_disp_out, __float_as_uint(1.0e20f)); }      // (the launches behind this one stay no-ops)
            }
            // one wave per tile: the bonded roles of this tile's atoms at x as well, partners through the same two rows; the force leaves
            // right away (three atomics per atom into the buffer this launch accumulates - nothing of this stays live across the chunk
            // loop).  (Eight waves per tile: the roles ride in extra workgroups of the launch, bonded_workgroup.)
            if (WPT == 1 && a.b_role_off) {
                const uint32_t rb = a.b_role_off[s], re = a.b_role_off[s + 1];
                if (re > rb) {
                    float bx = 0.f, by = 0.f, bz = 0.f;
                    RoleEnergies en;
                    for (uint32_t k = rb; k < re; ++k) role_eval_step<false>(a.b_roles[k], a.b_prm, x, a.posq, a.st_fprev, a.b_p, bx, by, bz, en);
                    float* const fb_ = reinterpret_cast<float*>(a.force) + (size_t)s * 4;
                    unsafeAtomicAdd(fb_, bx); unsafeAtomicAdd(fb_ + 1, by); unsafeAtomicAdd(fb_ + 2, bz);
                }
            }
        }
        s_xyzq[wave][lane] = x;
        WAVE_LDS_SYNC();
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float4 pi = s_xyzq[wave][ci * MDX_CLUSTER + ii];
            const float2 li = a.lj[t * MDX_TILE + ci * MDX_CLUSTER + ii];
            xi[ci] = pi.x; yi[ci] = pi.y; zi[ci] = pi.z; qi[ci] = pi.w; sgi[ci] = li.x; epi[ci] = li.y;
            fx[ci] = 0.f; fy[ci] = 0.f; fz[ci] = 0.f;
        }
        WAVE_LDS_SYNC();
    } else {
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const uint32_t s = t * MDX_TILE + ci * MDX_CLUSTER + ii;
        const float4 pi = a.posq[s];
        const float2 li = a.lj[s];
        xi[ci] = pi.x; yi[ci] = pi.y; zi[ci] = pi.z; qi[ci] = pi.w; sgi[ci] = li.x; epi[ci] = li.y;
        fx[ci] = 0.f; fy[ci] = 0.f; fz[ci] = 0.f;
    }
    }
    ListCounts cnt = a.counts[t_ok ? t : 0];
    if (!t_ok) { cnt.n_masked = 0; cnt.n_plain = 0; }
    const uint32_t e0 = a.entry_off[t_ok ? t : 0];
    // DUAL == 1: the inner list.  Same offsets and the same masked run as the Verlet list (exclusion masks are addressed
    // by chunk position); of the plain run, wave `part` finds the survivors of ITS chunks (c = part, part + WPT, ...)
    // compacted into the first of those chunk positions, and its own loop bound in inner_nch.
    const uint2* __restrict__ const entries = DUAL == 1 ? a.entries_in : a.entries;
    const uint32_t nmc = cnt.n_masked >> 3;
    const uint32_t nchunks = DUAL == 1 ? (t_ok ? a.inner_nch[t * 8 + part] : 0u) : (cnt.n_masked + cnt.n_plain) >> 3;
    const uint32_t mbase = a.mchunk_off[t_ok ? t : 0];
    float4* sx = s_xyzq[wave];
    float2* sl = s_lj[wave];
    double elj = 0.0, ecoul = 0.0, evir = 0.0, ecross = 0.0, edudl = 0.0;
    uint32_t own_bits = 0xFFu;   // ENERGY only: bit ci set <=> i-atom (ci, ii) is owned by this rank
    if (ENERGY && !a.energy_all) {
        own_bits = 0;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
            own_bits |= ((a.slot_flags[t * MDX_TILE + ci * MDX_CLUSTER + ii] >> 1) & 1u) << ci;
    }

    // two-deep software pipeline: entries of the wave's next-but-one chunk, atoms AND exclusion masks
    // of its next chunk are in flight while the current chunk is evaluated.  Everything a chunk
    // needs from HBM is issued one iteration ahead and consumed at the top of the loop, so the one
    // s_waitcnt vmcnt(0) per chunk sits where the data is a whole chunk old and the entry loop has
    // no vector-memory instruction in it (a mask fetched at its point of use - or an atomic issued
    // per entry - drags a vmcnt(0) into the loop and serialises the wave on memory latency).
    float4 nj = make_float4(0.f, 0.f, 0.f, 0.f), nf = nj;      // (STEP: nf = the force row beside the j-atom's Y)
    float2 nl = make_float2(0.f, 0.f);
    uint32_t ny = 13, njc = 0;
    float nown = 0.f;
    unsigned long long nmq = ~0ull;
    uint2 ent_n = make_uint2(0u, 13u);
    if ((uint32_t)part < nchunks) {
        const uint2 ent = entries[e0 + part * 8 + (lane >> 3)];
        if ((uint32_t)part < nmc) nmq = a.masks[(size_t)(mbase + part) * 64 + lane];
        if ((uint32_t)part + WPT < nchunks) ent_n = entries[e0 + (part + WPT) * 8 + (lane >> 3)];
        const uint32_t js = ent.x * MDX_CLUSTER + (lane & 7);
        nj = a.posq[js]; nl = a.lj[js]; ny = ent.y; njc = ent.x;
        if (STEP) nf = a.st_fprev[js];
        if (ENERGY && HALF) nown = a.energy_all ? 1.0f : (float)((a.slot_flags[js] >> 1) & 1u);
    }
    float* const fbase = reinterpret_cast<float*>(a.force);
    float4* const sg = s_g[HALF ? wave : 0];
    // Dual pair list.  Normal launch (DUAL 1): the list is the INNER one: cluster pairs that had an atom pair within
    // cutoff + inner_skin at the last pruning pass.  Pruning launch (DUAL 2; forced by the host after a rebuild, or asked
    // for by the drift pass through the step's prune word when some atom's path length since the last pass exceeded
    // inner_skin/2): walk the Verlet list, evaluate as usual, and while the distances are at hand ballot every cluster
    // pair against the inner radius; the surviving masks go to the inner list (masked run in place, this wave's plain
    // entries compacted), and the tile's path accumulators are cleared.  Decided on the device: the host enqueues blind.
    constexpr bool prune = DUAL == 2;
    uint32_t kept = 0;
#ifdef NB_HALF_STATS
    uint32_t dbg_jh = 0, dbg_ih = 0, dbg_q = 0, dbg_pairs = 0;
#endif
    // (pruning launch) first plain chunk of this wave's share, and how many compacted plain entries it has written
    const uint32_t c_first = nmc + ((uint32_t)part + WPT - nmc % WPT) % WPT;
    uint32_t wcur = 0;
    // HALF: the j-forces of a chunk collect in the wave's LDS strip and leave as three 64-lane
    // atomics (x, y, z of j-atom `lane`) when the chunk is done: always three, issued AFTER the
    // prefetch loads, so the loads' waits at the top of the next chunk are vmcnt(3) and the
    // atomics stay in flight (lanes of list padding add 0 to their own i-slot)
    // (HALF) retire the prologue's loads here: the loop then never waits at its top, where the
    // atomics of the previous chunk are still in flight
    if (HALF && NB_HALF_FLUSH) __builtin_amdgcn_s_waitcnt(0x0F70);
    const float rc2_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.p.rc2_lj)));   // an SGPR operand of v_cmpx
    // The chunk loop exists twice (round 3): the masked chunks at the head of the wave's share, then the plain ones - see the
    // entry loop.  The software pipeline's state (nj, nl, ny, njc, ent_n, nmq) carries over from the first loop into the second.
    auto chunk_pass = [&](auto masked_tag, const uint32_t c_begin, const uint32_t c_end) __attribute__((always_inline)) {
    constexpr bool MASKED_CHUNK = decltype(masked_tag)::value;
    for (uint32_t c = c_begin; c < c_end; c += WPT) {
        const uint32_t cur_y = ny, cur_jc = njc;
        constexpr bool masked = MASKED_CHUNK;
        if (masked) s_mask[wave][lane] = nmq;                   // read back a byte per entry: two VGPRs fewer
        if (ENERGY && HALF) s_ownj[wave][lane] = nown;
        if (HALF && NB_HALF_FLUSH) {
            // the lane that staged j-atom `lane` also flushes its force: remember where it goes.
            // List padding (imask 0) has no force: it will add 0 to this lane's own i-slot.
            const bool live = ((ny >> 8) & 0xFFu) != 0u;
            sg[lane] = make_float4(0.f, 0.f, 0.f, __uint_as_float(live ? njc * MDX_CLUSTER + (uint32_t)ii : t * MDX_TILE + lane));
        }
        {   // image shift, then park in LDS
            const uint32_t code = ny & 31u;
            const int kx = (int)(code % 3u) - 1, ky = (int)((code / 3u) % 3u) - 1, kz = (int)(code / 9u) - 1;
            if (STEP) nj = step_pos(nj, nf);
            nj.x += (float)kx * a.p.shift[0];
            nj.y += (float)ky * a.p.shift[1];
            nj.z += (float)kz * a.p.shift[2];
            sx[lane] = nj;
            sl[lane] = nl;
        }
        WAVE_LDS_SYNC();
        if (c + WPT < nmc) nmq = a.masks[(size_t)(mbase + c + WPT) * 64 + lane];
        if (c + WPT < nchunks) {
            const uint32_t js = ent_n.x * MDX_CLUSTER + (lane & 7);
            nj = a.posq[js]; nl = a.lj[js]; ny = ent_n.y; njc = ent_n.x;
            if (STEP) nf = a.st_fprev[js];
            if (ENERGY && HALF) nown = a.energy_all ? 1.0f : (float)((a.slot_flags[js] >> 1) & 1u);
            if (c + 2 * WPT < nchunks) ent_n = entries[e0 + (c + 2 * WPT) * 8 + (lane >> 3)];
        }
        float celj = 0.f, cecoul = 0.f, cevir = 0.f, cecross = 0.f, cedudl = 0.f;   // (ENERGY) fp32 partial sums of this chunk
        uint32_t newy = cur_y & 0xFFu;                                // (pruning launch) this lane's entry word with the inner mask
        // entry loop: each entry's j record is read from LDS where it is needed (any look-ahead measured slower).
        // Two copies of it (round 3, NB_SPLIT_MASKED): exclusion bits only exist in the masked chunks at the head of a tile's list;
        // the plain chunks (90-95 %) run a body without the per-pair v_bfe_i32 that turns a mask bit into the NaN addend of r^2
        // (one VALU instruction of ~29 per cluster pair; 16 k extra bytes of code).
#pragma unroll 1
        for (int e = 0; e < 8; ++e) {
            const float4 pj = sx[e * 8 + jj];
            const float2 lj = sl[e * 8 + jj];
            const uint32_t im = (__builtin_amdgcn_readlane(cur_y, e * 8) >> 8) & 0xFFu;  // wave-uniform
            if (im == 0) continue;
            uint32_t newm = 0;
            // (as EXCLUDED bits: bit ci set <=> this lane's pair with i-cluster ci is masked out)
            const int x8 = MASKED_CHUNK ? (int)(~(uint32_t)reinterpret_cast<const uint8_t*>(&s_mask[wave][lane])[e]) : 0;
            float g[3] = {0.f, 0.f, 0.f};
            float wj = 0.f;
            if (ENERGY && HALF) wj = 0.5f * s_ownj[wave][e * 8 + jj];
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (im & (1u << ci)) {
                    float e1 = 0.f, e2 = 0.f, e3 = 0.f, e4 = 0.f, e5 = 0.f;
                    // 0.0f or NaN: sign-extend bit ci of the exclusion byte over the word (one v_bfe_i32)
                    const float bias = MASKED_CHUNK ? __int_as_float((x8 << (31 - ci)) >> 31) : 0.f;
                    float r2v = 0.f;
                    constexpr bool ASM_PAIR = NB_ASM_PAIR && !ENERGY && COUL == CM_SHIFTED && !GEOM && SAMECUT && HALF && !ALCH;
                    if constexpr (ASM_PAIR)
                        pair_eval_asm<MASKED_CHUNK, prune>(xi[ci], yi[ci], zi[ci], qi[ci], sgi[ci], epi[ci], pj, lj, rc2_s, bias, fx[ci], fy[ci], fz[ci],
                                                           g[0], g[1], g[2], r2v);
                    else
                    pair_eval<ENERGY, COUL, GEOM, SAMECUT, true, HALF, ALCH, MASKED_CHUNK>(xi[ci], yi[ci], zi[ci], qi[ci], sgi[ci], epi[ci],
                                                                             pj, lj, true, a.p, fx[ci], fy[ci],
                                                                             fz[ci], e1, e2, g, ENERGY ? &e3 : nullptr,
                                                                             (ENERGY && ALCH) ? &e4 : nullptr, bias,
                                                                             prune ? &r2v : nullptr, (ENERGY && ALCH) ? &e5 : nullptr, etab);
                    if (prune) {   // any allowed atom pair of this cluster pair inside the inner radius?
                        const unsigned long long bb = __ballot(r2v < a.rin2);
                        if (bb != 0ull) newm |= 1u << ci;
#ifdef NB_HALF_STATS
                        if (bb != 0ull) {
                            const unsigned long long ilo = 0x0F0F0F0F0F0F0F0Full;
                            dbg_jh += ((uint32_t)bb != 0u) + ((bb >> 32) != 0ull);
                            dbg_ih += ((bb & ilo) != 0ull) + ((bb & ~ilo) != 0ull);
                            dbg_q += (((uint32_t)bb & 0x0F0F0F0Fu) != 0u) + (((uint32_t)bb & 0xF0F0F0F0u) != 0u) +
                                     (((uint32_t)(bb >> 32) & 0x0F0F0F0Fu) != 0u) + (((uint32_t)(bb >> 32) & 0xF0F0F0F0u) != 0u);
                            dbg_pairs += __popcll(bb);
                        }
#endif
                    }
                    if (ENERGY && HALF) {   // a pair's energy is split between the owners of its two atoms
                        const float w = wj + (((own_bits >> ci) & 1u) ? 0.5f : 0.f);
                        celj += w * e1; cecoul += w * e2; cevir += w * e3; cecross += w * e4; cedudl += w * e5;
                    } else if (ENERGY && ((own_bits >> ci) & 1u)) { celj += e1; cecoul += e2; cevir += e3; cecross += e4; cedudl += e5; }
                }
            }
            if (prune) {
                kept += __popc(newm);
                if (jj == e) newy |= newm << 8;
            }
            if (HALF) {
                // sum over the eight i-lanes of every j-atom: xor 1, xor 2 (quad_perm), then the
                // other quad of the row half (row_half_mirror) - every lane ends with the total
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    g[d] = dpp_xadd<0xB1>(g[d]);
                    g[d] = dpp_xadd<0x4E>(g[d]);
                    g[d] = dpp_xadd<0x141>(g[d]);
                }
                if (ii < 3) {
                    const float v = ii == 0 ? g[0] : (ii == 1 ? g[1] : g[2]);
                    if (NB_HALF_FLUSH) reinterpret_cast<float*>(sg)[(e * 8 + jj) * 4 + ii] = -v;
#ifdef NB_EXP_NOATOMIC
                    else if (v == 1.2345e30f) unsafeAtomicAdd(fbase + (size_t)__builtin_amdgcn_readlane(cur_jc, e * 8) * (MDX_CLUSTER * 4) + jj * 4 + ii, -v);
#else
                    else unsafeAtomicAdd(fbase + (size_t)__builtin_amdgcn_readlane(cur_jc, e * 8) * (MDX_CLUSTER * 4) + jj * 4 + ii, -v);
#endif
                }
            }
        }
        if (ENERGY) { elj += (double)celj; ecoul += (double)cecoul; evir += (double)cevir; ecross += (double)cecross; edudl += (double)cedudl; }
        if (prune) {
            if (masked) {                       // exclusion masks are addressed by chunk position: the entry stays where it is
                if (ii == 0) a.entries_in[e0 + c * 8 + jj] = make_uint2(cur_jc, newy);
            } else {                            // plain run: survivors move up into this wave's first chunk positions
                const bool alive = ii == 0 && (newy >> 8) != 0u;
                const unsigned long long bal = __ballot(alive);
                if (alive) {
                    const uint32_t k = wcur + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    a.entries_in[e0 + (c_first + (k >> 3) * WPT) * 8 + (k & 7u)] = make_uint2(cur_jc, newy);
                }
                wcur += (uint32_t)__popcll(bal);
            }
        }
        WAVE_LDS_SYNC();
        if (HALF && NB_HALF_FLUSH) {
            // Flush: 192 floats = 3 instructions x 64 lanes, lane l of instruction k taking float
            // k*64 + l of the packed [64 atoms][xyz] view, so consecutive lanes hit consecutive words
            // and an instruction touches each 128-B line once.  (x of all atoms, then y, then z hits
            // every line three times in a row and is 3x slower; a per-entry 24-lane atomic inside the
            // entry loop makes every wait in the loop a vmcnt(0): tools/ubench/atomic_flush.hip.)
            // The prefetch issued before this chunk's arithmetic has landed long ago: retire it here,
            // so that nothing after the atomics has to wait on the vector-memory counter.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
            const float* const sgf = reinterpret_cast<const float*>(sg);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t fidx = (uint32_t)(k * 64 + lane), atom = (fidx * 171u) >> 9, comp = fidx - 3u * atom;
                const uint32_t slot = __float_as_uint(sgf[atom * 4 + 3]);
                unsafeAtomicAdd(fbase + (size_t)slot * 4 + comp, sgf[atom * 4 + comp]);
            }
        }
    }
    };
#ifndef NB_SPLIT_MASKED
#define NB_SPLIT_MASKED 1
#endif
    if (NB_SPLIT_MASKED) {
        const uint32_t c_mid = c_first < nchunks ? c_first : nchunks;      // this wave's first plain chunk
        chunk_pass(std::true_type{}, (uint32_t)part, c_mid);
        chunk_pass(std::false_type{}, c_mid, nchunks);
    } else chunk_pass(std::true_type{}, (uint32_t)part, nchunks);
    // sum the eight j-lanes of every i-atom (lanes ii, ii+8, ..., ii+56), then lane (ii, jj) keeps
    // i-cluster ci == jj: slot tile*64 + jj*8 + ii == tile*64 + lane, a coalesced store.
    float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        float x = fx[ci], y = fy[ci], z = fz[ci];
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            x += __shfl_xor(x, m); y += __shfl_xor(y, m); z += __shfl_xor(z, m);
        }
        if (jj == ci) { ox = x; oy = y; oz = z; }
    }
    float* const fi = fbase + (size_t)(t * MDX_TILE + lane) * 4;
    if (prune && t_ok) {
        // pad the last compacted chunk with null entries (imask 0), publish this wave's loop bound
        const uint32_t nfull = (wcur + 7u) >> 3;
        if ((uint32_t)lane < nfull * 8u - wcur) {
            const uint32_t k = wcur + (uint32_t)lane;
            a.entries_in[e0 + (c_first + (k >> 3) * WPT) * 8 + (k & 7u)] = make_uint2(a.T * MDX_CL_PER_TILE, 13u);
        }
        if (lane == 0) a.inner_nch[t * 8 + part] = nfull ? c_first + (nfull - 1u) * WPT + 1u : nmc;
    }
    if (prune) {
        // path lengths count from this pass (a boundary pass asked for by the ghosts alone restarts the ghosts only)
        if (!STEP && part == 0 && t_ok && (owned_prune || !(a.slot_flags[t * MDX_TILE + lane] & 2u))) {      // (STEP: the tile's wave did this where it had x)
            const uint32_t s = t * MDX_TILE + lane;
            if (a.path) {      // path split: the accumulator restarts, and the displacement reached so far is kept for the drift pass's bound
                const float4 x = a.posq[s], r = a.ref[s];
                const float dx = x.x - r.x, dy = x.y - r.y, dz = x.z - r.z;
                a.path[s] = 0.f; a.dprune[s] = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
            } else a.ref[s].w = 0.f;
        }
        if (lane == 0 && kept) atomicAdd(a.inner_count + ((blk * BW + wave) & (MDX_EPART - 1)), (unsigned long long)kept);
        if (lane == 0 && blk == 0 && wave == 0) atomicAdd(a.inner_count + MDX_EPART, 1ull);
#ifdef NB_HALF_STATS
        if (lane == 0) {
            atomicAdd(a.inner_count + MDX_EPART + 1, (unsigned long long)dbg_jh); atomicAdd(a.inner_count + MDX_EPART + 2, (unsigned long long)dbg_ih);
            atomicAdd(a.inner_count + MDX_EPART + 3, (unsigned long long)dbg_q); atomicAdd(a.inner_count + MDX_EPART + 4, (unsigned long long)kept);
            atomicAdd(a.inner_count + MDX_EPART + 5, (unsigned long long)dbg_pairs);
        }
#endif
    }
    if (WPT > 1) {   // fixed-order sum of the waves' partial forces
        s_red[wave][0][lane] = ox; s_red[wave][1][lane] = oy; s_red[wave][2][lane] = oz;
        __syncthreads();
        if ((SPLIT > 1 ? wave == 0 : part == 0) && t_ok) {
            const int w0 = SPLIT > 1 ? 0 : tib * WPT;
            ox = s_red[w0][0][lane]; oy = s_red[w0][1][lane]; oz = s_red[w0][2][lane];
#pragma unroll
            for (int w = 1; w < (SPLIT > 1 ? BW : WPT); ++w) { ox += s_red[w0 + w][0][lane]; oy += s_red[w0 + w][1][lane]; oz += s_red[w0 + w][2][lane]; }
            if (HALF) { unsafeAtomicAdd(fi, ox); unsafeAtomicAdd(fi + 1, oy); unsafeAtomicAdd(fi + 2, oz); }
            else a.force[t * MDX_TILE + lane] = make_float4(ox, oy, oz, 0.f);
        }
    } else {
        if (HALF) { unsafeAtomicAdd(fi, ox); unsafeAtomicAdd(fi + 1, oy); unsafeAtomicAdd(fi + 2, oz); }
        else a.force[t * MDX_TILE + lane] = make_float4(ox, oy, oz, 0.f);
    }
    if (ENERGY) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            elj += __shfl_xor(elj, m);
            ecoul += __shfl_xor(ecoul, m);
            evir += __shfl_xor(evir, m);
            if (ALCH) { ecross += __shfl_xor(ecross, m); edudl += __shfl_xor(edudl, m); }
        }
        if (lane == 0) {   // full list: every pair is seen from both sides.  The atomics are spread over MDX_EPART
                           // slots: 65 k waves adding into three words took 2 ms (contended f64 atomics, ~10 ns each)
            double* q = a.energy + EN_COUNT + 8 + MDX_ESTRIDE * ((blk * BW + wave) & (MDX_EPART - 1));
            atomicAdd(q, HALF ? elj : 0.5 * elj);
            atomicAdd(q + 1, HALF ? ecoul : 0.5 * ecoul);
            atomicAdd(q + 2, HALF ? evir : 0.5 * evir);
            if (ALCH) { atomicAdd(q + 3, HALF ? ecross : 0.5 * ecross); atomicAdd(q + 4, HALF ? edudl : 0.5 * edudl); }
        }
    }
}

// The bonded gather as extra workgroups of the pair launch (DUAL 4, systems of a few hundred tiles): the pair tiles of
// such a system fill a fraction of the chip, so the gather's waves run beside them on idle CUs and the step is one
// launch shorter - at 23 k atoms the separate gather was 9.7 us of kernel and a ~4 us launch gap in a 72 us step.  Four
// lanes per atom as in bonded_gather_kernel (mdx_bonded.hip); the force leaves through atomics, as the pair kernel's
// does.  (At 1 M atoms the same arrangement costs more than it hides - the two compete for the memory-side atomic
// path, DESIGN.md section 4 - so the large-system classes keep the separate launch.)
template <bool STEP = false>
__device__ __forceinline__ void bonded_workgroup(const NbArgs& a, uint32_t wg, uint32_t wg_threads) {
    const uint32_t tid = wg * wg_threads + threadIdx.x;
    const uint32_t s = tid >> 2, q4 = tid & 3u;
    if (s >= a.b_S) return;
    const uint32_t rb = a.b_role_off[s], re = a.b_role_off[s + 1];
    if (re <= rb) return;      // quad-uniform
    // (one launch per step: positions in the step form, every atom reconstructed from its two rows - mdx_bonded_dev.h step_pos)
    const float4 self = STEP ? step_pos(a.posq[s], a.st_fprev[s]) : a.posq[s];
    float fx = 0.f, fy = 0.f, fz = 0.f;
    RoleEnergies en;
    for (uint32_t k = rb + q4; k < re; k += 4) {
        const RoleRec r = a.b_roles[k];
        if (STEP) role_eval_step<false>(r, a.b_prm, self, a.posq, a.st_fprev, a.b_p, fx, fy, fz, en);
        else role_eval<false>(r, a.b_prm, self, a.posq, a.b_p, fx, fy, fz, en);
    }
    fx = dpp_xadd<0xB1>(fx); fy = dpp_xadd<0xB1>(fy); fz = dpp_xadd<0xB1>(fz);   // lane ^ 1
    fx = dpp_xadd<0x4E>(fx); fy = dpp_xadd<0x4E>(fy); fz = dpp_xadd<0x4E>(fz);   // lane ^ 2
    if (q4 < 3) {
        float* const f = reinterpret_cast<float*>(a.force) + (size_t)s * 4;
        unsafeAtomicAdd(f + q4, q4 == 0 ? fx : (q4 == 1 ? fy : fz));
    }
}

#ifndef NB_STEP_WAVES3
#define NB_STEP_WAVES3 0      // one launch per step: 1 = three waves per SIMD (168 VGPRs) instead of four (128)
#endif
#ifndef NB_MERGED_NOINLINE
#define NB_MERGED_NOINLINE 0
#endif
template <bool ENERGY, int COUL, bool GEOM, bool SAMECUT, int WPT, bool HALF, bool ALCH, int DUAL, int BW>
__device__ __attribute__((noinline)) void nb_cluster_body_call(const NbArgs& a, const bool owned_prune, float4 (*s_xyzq)[64], float2 (*s_lj)[64],
                                                               float (*s_red)[3][64], float (*s_ownj)[64], float4 (*s_g)[64],
                                                               unsigned long long (*s_mask)[64], const float4* __restrict__ etab = nullptr) {
    nb_cluster_body<ENERGY, COUL, GEOM, SAMECUT, WPT, HALF, ALCH, DUAL, BW>(a, owned_prune, s_xyzq, s_lj, s_red, s_ownj, s_g, s_mask, etab);
}

template <bool ENERGY, int COUL, bool GEOM, bool SAMECUT, int WPT, bool HALF, bool ALCH = false, int DUAL = 0, bool FB = false, int SPLIT = 1>
__global__ __launch_bounds__((SPLIT > 1 ? WPT / SPLIT : (WPT > NB_WAVES ? WPT : NB_WAVES)) * 64, ((HALF && NB_HALF_FLUSH) || ENERGY || (DUAL == 5 && NB_STEP_WAVES3)) ? 3 : 4) void nb_cluster_kernel(NbArgs a) {
    if (a.gate && *a.gate > a.thr_bits) {
        // (one launch per step: the stale word travels on, the launches behind this one stay no-ops too)
        if ((DUAL == 5 || DUAL == 6) && a.st_disp_out && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(a.st_disp_out, *a.gate);
        return;
    }
    // one launch per step: DUAL 5 = the merged launch of the one-wave class, the tile's wave evaluates its atoms' roles; DUAL 6 = the merged
    // launch of the eight-waves class with the bonded gather in extra workgroups (as DUAL 4)
    static_assert(DUAL != 5 || (WPT == 1 && !FB && SPLIT == 1), "one launch per step: the one-wave-per-tile class");
    static_assert(DUAL != 6 || (WPT == 8 && !FB), "one launch per step: the eight-waves-per-tile class");
    constexpr bool STEP = DUAL == 5 || DUAL == 6;
    static_assert(DUAL == 0 || (HALF && !ENERGY), "the dual list exists for the half-list force kernel");
    static_assert(!FB || DUAL != 0, "the bonded workgroups ride with the dual-list launches");
    constexpr int BW = SPLIT > 1 ? WPT / SPLIT : (WPT > NB_WAVES ? WPT : NB_WAVES);       // waves per workgroup
    if ((DUAL == 4 || DUAL == 6 || FB) && blockIdx.x >= a.pair_grid) {
        // (a twin launch: the bonded workgroups run in whichever of the two the device executes)
        if (DUAL == 1 || DUAL == 2) {
            const bool want_prune = (a.force_prune | *a.prune_flag) != 0u || (a.prune_flag2 && *a.prune_flag2 != 0u);
            if (want_prune != (DUAL == 2)) return;
        }
        bonded_workgroup<STEP>(a, blockIdx.x - a.pair_grid, BW * 64);
        return;
    }
    __shared__ float4 s_xyzq[BW][64];
    __shared__ float2 s_lj[BW][64];
    __shared__ float s_red[WPT > 1 ? BW : 1][3][64];
    __shared__ float s_ownj[(ENERGY && HALF) ? BW : 1][64];   // 1.0 <=> the staged j-atom is owned here
    __shared__ float4 s_g[HALF ? BW : 1][64];                 // minus the chunk's j-forces (.w: the j-slot), until flushed
    __shared__ unsigned long long s_mask[BW][64];             // a masked chunk's per-lane exclusion bits
    // Ewald force flavour: the table of the smooth part (mdx_pair_dev.h) in dynamic LDS, one copy per workgroup
    extern __shared__ float4 s_dyn_etab[];
    const float4* etab = nullptr;
    if (COUL == CM_EWALD_TAB && !ENERGY && !ALCH) {
        for (uint32_t k = threadIdx.x; k < a.p.etab_n; k += BW * 64) s_dyn_etab[k] = a.p.etab[k];
        __syncthreads();
        // (biased by the byte offset of the table's first binade: pair_eval adds the masked bits of x, shifted, and nothing else)
        etab = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_dyn_etab) - ((EWALD_TAB_CBITS >> a.p.etab_shift) << 4));
    }
    if (DUAL == 0) {
        nb_cluster_body<ENERGY, COUL, GEOM, SAMECUT, WPT, HALF, ALCH, 0, BW>(a, true, s_xyzq, s_lj, s_red, s_ownj, s_g, s_mask, etab);
    } else {
        const bool owned_prune = (a.force_prune | *a.prune_flag) != 0u;
        const bool want_prune = owned_prune || (a.prune_flag2 && *a.prune_flag2 != 0u);
        if (DUAL >= 3) {        // merged launch: the device picks the body
#if NB_MERGED_NOINLINE
            // (experiment: the two bodies as real functions, so that each keeps the register allocation of its stand-alone kernel)
            if (want_prune) nb_cluster_body_call<ENERGY, COUL, GEOM, SAMECUT, WPT, HALF, ALCH, 2, BW>(a, owned_prune, s_xyzq, s_lj, s_red, s_ownj, s_g, s_mask, etab);
            else nb_cluster_body_call<ENERGY, COUL, GEOM, SAMECUT, WPT, HALF, ALCH, 1, BW>(a, owned_prune, s_xyzq, s_lj, s_red, s_ownj, s_g, s_mask, etab);
#else
            if (want_prune) nb_cluster_body<ENERGY, COUL, GEOM, SAMECUT, WPT, HALF, ALCH, 2, BW, STEP>(a, owned_prune, s_xyzq, s_lj, s_red, s_ownj, s_g, s_mask, etab);
            else nb_cluster_body<ENERGY, COUL, GEOM, SAMECUT, WPT, HALF, ALCH, 1, BW, STEP>(a, owned_prune, s_xyzq, s_lj, s_red, s_ownj, s_g, s_mask, etab);
#endif
        } else {
            if (want_prune != (DUAL == 2)) return;
            nb_cluster_body<ENERGY, COUL, GEOM, SAMECUT, WPT, HALF, ALCH, DUAL == 2 ? 2 : 1, BW>(a, owned_prune, s_xyzq, s_lj, s_red, s_ownj, s_g, s_mask, etab);
        }
    }
}

template <bool ENERGY, int COUL>
void launch_variant(mdx_handle* h, const NbArgs& a, bool geom, bool samecut) {
    const int var = mdx_nb_variant(h);
    // waves per tile: split a tile's list over the 4 waves of its workgroup when the launch has
    // fewer tiles than ~3 per SIMD (the chip holds 4 waves/SIMD of this kernel on 1024 SIMDs)
    int wpt = 1;
    const bool half = var == 5;
    // measured: 4 beats 1 at every size, 8 below ~130 k atoms (round 3: 192 k atoms / 3.1 k tiles 0.137 ms with 8, 0.125 with 4, 0.127
    // with 2; one rank of an 8-rank water1M - 2.0 k owned + 1.3 k ghost tiles - 0.111 / 0.106 / 0.112)
    // with 2; one rank of an 8-rank water1M - 2.0 k owned + 1.3 k ghost tiles - 0.111 with the bonded gather inside / 0.106 + 0.010 for its
    // own launch / 0.112: a decomposed handle, whose step loop cannot fuse the gather into the drift pass, keeps 8 up to 4096 tiles)
    if (var == 2) wpt = (a.T < mdx_wpt8_below(h)) ? 8 : 4;
    if (half) wpt = mdx_nb_wpt_half(h, a.T);               // (shared with the tile order of the list rebuild: mdx_internal.h)
    if (var == 4) wpt = 4;
    const uint32_t bw = std::max(wpt, NB_WAVES);
    const uint32_t tpb = (var == 1) ? NB_WAVES : bw / wpt;
    const uint32_t nblocks = ((a.tile_order ? a.t_count : a.T) + tpb - 1) / tpb;
    const uint32_t grid = ((nblocks + 7) / 8) * 8;
    if (nblocks == 0) return;
    dim3 g(grid), b(bw * 64);
    // MDX_NB_LDS_PAD=<bytes> (experiment): dynamic LDS the pair kernel asks for and never touches, on handles whose reciprocal-space
    // chain runs on a side stream.  Four waves of this kernel per SIMD hold the whole register file, so the chain's kernels only get
    // a CU when a pair workgroup retires - the two time-slice instead of overlapping; padding the LDS footprint caps the pair
    // workgroups per CU and leaves registers and LDS for the chain.
    static const uint32_t lds_pad_env = [] { const char* e = std::getenv("MDX_NB_LDS_PAD"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    const uint32_t lds_pad = ((h->pme_on && h->pme_overlap && !ENERGY) ? lds_pad_env : 0u) +
                             ((COUL == CM_EWALD_TAB && !ENERGY) ? a.p.etab_n * (uint32_t)sizeof(float4) : 0u);      // + the Ewald table
    // ONE kernel picks the inner-walk or the pruning body on the device (round 2 measured the merged launch at +1 % for 23 k atoms
    // and -0.4 % at 1 M, and launched the two flavours back to back for the large classes, the device running exactly one; since
    // the chunk loop exists twice the merged kernel is faster there too: the gated-off twin was ~5 us of a 550 us step).
    // MDX_DUAL_MERGED=0: never merge; 1: the eight-waves-per-tile class only.
    static const bool dual_merged = [] { const char* e = std::getenv("MDX_DUAL_MERGED"); return !(e && e[0] == '0'); }();
    // (round 3, after the two chunk loops: merged wins at every size - water1M 1812 -> 1827 steps/s; MDX_DUAL_MERGED=1: small class only)
    static const bool merge_all = [] { const char* e = std::getenv("MDX_DUAL_MERGED"); return !(e && (e[0] == '0' || e[0] == '1')); }();
    // below ~1000 tiles a tile's eight waves go to two workgroups of four (MDX_TILE_SPLIT=0 / 1 forces)
    static const int split_env = [] { const char* e = std::getenv("MDX_TILE_SPLIT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    const bool split2 = split_env >= 0 ? split_env == 1 : (a.tile_order ? a.t_count : a.T) < 1024u;
    static const bool fb_all = [] { const char* e = std::getenv("MDX_FUSE_BONDED"); return e && e[0] == '2'; }();   // A/B: also the twin launches of the large classes
    // which instantiation goes out (mdx_pair_launch_info: the parity tests name the body they hold against the oracle)
    auto note = [&](int dual, int w, int split, int fb) {
        uint32_t* const o = a.inner ? h->pair_info_step : h->pair_info_any;
        o[0] = (uint32_t)w; o[1] = (uint32_t)dual; o[2] = half ? 1u : 0u; o[3] = (uint32_t)COUL; o[4] = ENERGY ? 1u : 0u;
        o[5] = (uint32_t)split; o[6] = (uint32_t)fb; o[7] = a.tile_order ? a.t_count : a.T;
    };
    // dual list: the inner-walk kernel and the pruning kernel back to back, the device runs exactly one of them
#define NB_STEP(G, S)                                                                                                 \
    do {                                                                                                               \
        if constexpr (!ENERGY && !G && COUL != CM_SOFT) {                                                              \
            if (wpt == 1 && half && a.inner) {                                                                         \
                hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, false, S, 1, true, false, 5>), g, b, lds_pad, h->stream, a); \
                note(5, 1, 1, 0);                                                                                      \
            } else if (wpt == 8 && half && a.inner && split2) {   /* a tile = two workgroups of four waves */          \
                NbArgs af = a;                                                                                         \
                const uint32_t nb2 = (a.tile_order ? a.t_count : a.T) * 2u;                                            \
                af.pair_grid = ((nb2 + 7) / 8) * 8;                                                                    \
                const dim3 gf(af.pair_grid + (a.b_S ? (uint32_t)(((size_t)a.b_S * 4 + 255) / 256) : 0u)), bf(256);     \
                hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, false, S, 8, true, false, 6, false, 2>), gf, bf, lds_pad, h->stream, af); \
                note(6, 8, 2, 0);                                                                                      \
            } else if (wpt == 8 && half && a.inner) {                                                                  \
                NbArgs af = a; af.pair_grid = grid;                                                                    \
                const dim3 gf(grid + (a.b_S ? (uint32_t)(((size_t)a.b_S * 4 + bw * 64 - 1) / (bw * 64)) : 0u));        \
                hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, false, S, 8, true, false, 6>), gf, b, lds_pad, h->stream, af); \
                note(6, 8, 1, 0);                                                                                      \
            } else h->onepass_refused = true;                                                                          \
        } else h->onepass_refused = true;                                                                              \
    } while (0)
#define NB_DUAL(G, S, D)                                                                                              \
    do {                                                                                                               \
        if (ENERGY) break;                                                                                             \
        if (fb_all && a.b_S && (wpt == 2 || wpt == 4)) {   /* bonded workgroups behind the pair grid of both twins */  \
            NbArgs af = a; af.pair_grid = grid;                                                                        \
            const dim3 gf(grid + (uint32_t)(((size_t)a.b_S * 4 + bw * 64 - 1) / (bw * 64)));                           \
            if (wpt == 2) hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 2, true, false, D, true>), gf, b, lds_pad, h->stream, af); \
            else hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 4, true, false, D, true>), gf, b, lds_pad, h->stream, af);          \
            h->bonded_fused = true; note(D, wpt, 1, 1);                                                                \
            break;                                                                                                     \
        }                                                                                                              \
        note(D, wpt == 8 || wpt == 2 || wpt == 1 ? wpt : 4, 1, 0);                                                     \
        if (wpt == 8) hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 8, true, false, D>), g, b, lds_pad, h->stream, a); \
        else if (wpt == 2) hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 2, true, false, D>), g, b, lds_pad, h->stream, a); \
        else if (wpt == 1) hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 1, true, false, D>), g, b, lds_pad, h->stream, a); \
        else hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 4, true, false, D>), g, b, lds_pad, h->stream, a);               \
    } while (0)
#define NB_LAUNCH(G, S)                                                                                    \
    do {                                                                                                   \
        note(0, var == 1 ? 0 : ((h->alch_on && wpt != 8) ? 4 : wpt), 1, 0);   /* (the dual-list branches below overwrite it) */ \
        if (h->alch_on && wpt == 8) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 8, true, COUL != CM_EWALD_TAB>), g, b, lds_pad, h->stream, a); \
        else if (h->alch_on) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 4, true, COUL != CM_EWALD_TAB>), g, b, lds_pad, h->stream, a); \
        else if (a.st_fprev) { NB_STEP(G, S); }                                                                    \
        else if (var == 1) hipLaunchKernelGGL((nb_tile_kernel<ENERGY, COUL, G, S>), g, b, lds_pad, h->stream, a);     \
        else if (half && a.inner && dual_merged && wpt == 8 && split2 && !ENERGY) {   /* a tile = two workgroups of four waves */ \
            NbArgs af = a;                                                                                         \
            const uint32_t nb2 = (a.tile_order ? a.t_count : a.T) * 2u;                                            \
            af.pair_grid = ((nb2 + 7) / 8) * 8;                                                                    \
            const dim3 gf(af.pair_grid + (a.b_S ? (uint32_t)(((size_t)a.b_S * 4 + 255) / 256) : 0u)), bf(256);     \
            if (a.b_S) { hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 8, true, false, 4, false, 2>), gf, bf, lds_pad, h->stream, af); h->bonded_fused = true; note(4, 8, 2, 0); } \
            else { hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 8, true, false, 3, false, 2>), gf, bf, lds_pad, h->stream, af); note(3, 8, 2, 0); } \
        }                                                                                                          \
        else if (half && a.inner && dual_merged && wpt == 8 && a.b_S && !ENERGY) {                                 \
            NbArgs af = a; af.pair_grid = grid;                                                                    \
            const dim3 gf(grid + (uint32_t)(((size_t)a.b_S * 4 + bw * 64 - 1) / (bw * 64)));                       \
            hipLaunchKernelGGL((nb_cluster_kernel<false, COUL, G, S, 8, true, false, 4>), gf, b, lds_pad, h->stream, af); \
            h->bonded_fused = true; note(4, 8, 1, 0);                                                              \
        }                                                                                                          \
        else if (half && a.inner && dual_merged && (wpt == 8 || merge_all)) { NB_DUAL(G, S, 3); }                  \
        else if (half && a.inner) { if (!a.force_prune) NB_DUAL(G, S, 1); NB_DUAL(G, S, 2); }   /* (a pass the host forces: no twin) */ \
        else if (half && wpt == 8) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 8, true>), g, b, lds_pad, h->stream, a); \
        else if (half && wpt == 2) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 2, true>), g, b, lds_pad, h->stream, a); \
        else if (half && wpt == 1) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 1, true>), g, b, lds_pad, h->stream, a); \
        else if (half) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 4, true>), g, b, lds_pad, h->stream, a); \
        else if (wpt == 8) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 8, false>), g, b, lds_pad, h->stream, a); \
        else if (wpt == 4) hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 4, false>), g, b, lds_pad, h->stream, a); \
        else hipLaunchKernelGGL((nb_cluster_kernel<ENERGY, COUL, G, S, 1, false>), g, b, lds_pad, h->stream, a);        \
    } while (0)
    if (geom) { if (samecut) NB_LAUNCH(true, true); else NB_LAUNCH(true, false); }
    else      { if (samecut) NB_LAUNCH(false, true); else NB_LAUNCH(false, false); }
#undef NB_LAUNCH
#undef NB_DUAL
#undef NB_STEP
}

