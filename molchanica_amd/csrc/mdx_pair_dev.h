// mdx_pair_dev.h - the argument block of the pair kernels and the evaluation of ONE pair (Lennard-Jones 12-6 + Coulomb in its four
// treatments), shared by the pair kernels of the step loop (mdx_nonbonded.hip) and the per-group energy matrix (mdx_groups.hip).
// The formulas follow the only in-tree statement of them, /root/reference src/cuda/util.cu:53-63 (Coulomb, dir = tgt - src) and
// :92-140 (LJ 12-6: F = dir*24 eps (2 s^12 - s^6)/r, E = 4 eps (s^12 - s^6)).
#pragma once
#include "mdx_internal.h"

struct NbParams;
// cutoffs, image shifts and the Coulomb treatment of a handle as the pair kernels take them; *mode_out: CM_* below (mdx_nonbonded.hip)
void mdx_fill_nb_params(const mdx_handle* h, NbParams& p, int* mode_out, bool* geom_out, bool* samecut_out);

#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

struct NbArgs {
    uint32_t T;
    const float4* posq; const float2* lj;
    const ListCounts* counts; const uint32_t* entry_off; const uint32_t* mchunk_off;
    const uint2* entries; const unsigned long long* masks;
    float4* force; double* energy;
    const uint8_t* slot_flags;   // bit1: owned (energy of a ghost i-atom belongs to its owner's rank)
    NbParams p;
    const uint32_t* gate; uint32_t thr_bits;
    // dual pair list (half-list force kernel of the step loop only; `inner` = 0 elsewhere)
    uint32_t inner;               // 1: walk the inner list unless this launch prunes
    uint32_t force_prune;         // 1: this launch prunes regardless of the flag word
    const uint32_t* prune_flag;   // ctl.prune[step + 1], raised by the drift pass
    float rin2;                   // (cutoff + inner_skin)^2
    uint2* entries_in;            // the inner list: same layout as `entries`, every wave's share of the plain run compacted
    uint32_t* inner_nch;          // [T * 8 + part] chunk-loop bound of wave `part` of a tile in the inner list
    float4* ref; unsigned long long* inner_count;
    float* path; float* dprune;   // path split (mdx_internal.h): the pruning pass clears path[] and stores |x - ref| in dprune[]; null: ref[].w
    uint32_t xcd_interleave;      // 1: workgroup b takes tile group b (round-robin over the XCDs) instead of a contiguous eighth per XCD
    // decomposed handle, interior / boundary split: this launch covers tiles tile_order[t_first .. t_first + t_count)
    const uint32_t* tile_order; uint32_t t_first, t_count;
    const uint32_t* prune_flag2;  // boundary launch: the ghosts' prune word (raised by the halo unpack); then only a pass asked for by
                                  // the owned atoms' word clears the owned atoms' path accumulators (the interior lists depend on them)
    // small systems (DUAL 4): workgroups pair_grid .. of the launch evaluate the bonded gather beside the pair tiles
    uint32_t energy_all;          // half-shell decomposition: every pair this rank evaluates is evaluated nowhere else - full weight
    uint32_t pair_grid, b_S;
    const uint32_t* b_role_off; const RoleRec* b_roles; const float4* b_prm;
    BondedParams b_p;
    // One launch per step (round 6; DUAL 5, one wave per tile, single-device handles without constraints: mdx_step "onepass").  posq = Y
    // (mdx_bonded_dev.h step_pos), st_fprev = the complete forces of the previous position stage with .w = dt^2 418.4/m, force = the buffer
    // this launch accumulates into (zeroed by the launch before it).  The tile's wave first finishes the last step for its 64 atoms -
    // x = Y + w F, full kick, next Y - then evaluates pairs and its atoms' bonded roles (b_role_off / b_roles / b_prm / b_p) at x;
    // j-atoms and bonded partners are reconstructed from the same two rows.  null st_fprev: not this flavour.
    const float4* st_fprev; float4* st_yout; float4* st_fnext; float4* st_vel;
    float st_dt, st_path_thr;
    uint32_t st_last;             // 1: the last launch of a chunk writes x itself (no pre-drift) and raises no words
    float st_grant;               // scale of the kick the words grant (1; tests make it negative to drive the way back)
    float st_kick;                // 0.5: the first launch of a chunk finishes a step that opens with a half kick (its force rows carry w / 2); else 1
    uint32_t* st_disp_out;        // ctl.disp2[s + 2]: (bound of |x - ref| at the NEXT position stage)^2, raised only above thr_bits
    uint32_t* st_prune_out;       // ctl.prune[s + 2]
    uint32_t* st_viol;            // ctl.viol[s]: the stage this launch found contradicts the words that let it run - the host takes the step back to a list rebuild
};

enum { CM_SHIFTED = 0, CM_RF = 1, CM_EWALD = 2, CM_SOFT = 3,
       CM_EWALD_TAB = 4 };     // Ewald real space with the smooth part from the LDS table: force-only, non-alchemical launches (mdx_launch_nonbonded picks it)

// Table of the smooth part of the Ewald real-space force (round 5).  erfc(beta r)/r^3 + 2 beta/sqrt(pi) exp(-beta^2 r^2)/r^2 =
// 1/r^3 - g(r^2) with g = [erf(beta r)/r - 2 beta/sqrt(pi) exp(-beta^2 r^2)] / r^2: bounded (4 beta^3 / (3 sqrt(pi)) at r = 0) and
// smooth.  The table is indexed by the FLOAT BITS of x = r^2 s + EWALD_TAB_C, s = max(1, (beta / 0.3)^2) - six mantissa bits per octave
// (seven above beta = 0.36) - and holds a PARABOLA per interval (through g at both ends and the middle): g = a + dx (b + dx c), dx = x -
// node, node = x with the low bits cleared.  ~300 entries of 16 bytes (4.8 kB, staged in LDS by every workgroup).  Max |g error| 2e-9 at
// beta 0.3 (fp32 evaluation included): 2e-6 kcal/mol/A on the strongest pair of OPC water at any distance - below the erfc formula it replaces (|error|
// 1.5e-7 in erfc) - growing with beta^3 only (s keeps the resolution in beta^2 r^2).  (Straight lines need ~2000 entries for that: the
// 1/r^3 tail of g keeps its curvature; 600 log-spaced lines left 7e-5 per pair.)  Per pair: three fma, two ands, shift, two subs, one
// ds_read_b128 - against two quarter-rate transcendentals (v_exp_f32, v_rcp_f32) and 12 VALU of the closed form.  Energies keep the
// closed form (they need erfc itself).
#define EWALD_TAB_C 4.0f
#define EWALD_TAB_CBITS 0x40800000u      // the bits of 4.0f
#define EWALD_TAB_MAX 1024u

// One pair, seen from atom i: adds the force on i.  `allowed` carries the exclusion mask bit.
// BRANCHY = true wraps everything behind the cutoff test in a divergent `if`: hipcc emits an
// exec-masked region with an s_cbranch_execz early-out, so a cluster pair with no lane inside the
// cutoff costs 7 VALU ops instead of 25 (the cluster kernel: ~40 % of its cluster pairs).  The
// whole-tile kernel keeps the straight-line select form, which the compiler can software-pipeline
// across its unrolled j loop.
//
// ALCH = true: thermodynamic-integration window.  The atoms of the coupled molecule carry a negative (or -0.0)
// sqrt(24 eps); the sign of the product eps_i * eps_j is then the "exactly one of the two is coupled" flag, and
// such a pair's force and energy are scaled by p.alch_scale = 1 - lambda (3 VALU ops; a separate instantiation,
// the default kernels do not pay for it).  *ecross collects the UNSCALED energy of those pairs: dU/dlambda = -it.
//
// NANMASK = true (the cluster kernels): the exclusion bit arrives as `r2bias` = 0.0f (allowed) or NaN (excluded) and
// is the addend of the first FMA of r^2, so an excluded pair fails every cutoff comparison by itself: one VALU op
// (v_bfe_i32 of the mask byte) instead of v_and + v_cmp + s_and, and one step less in the dependent chain in front
// of the exec-mask branch.  Adding 0.0f is exact: r^2 has the bits of the plain expression.  Requires BRANCHY (the
// NaN must never reach an accumulator).
template <bool ENERGY, int COUL, bool GEOM, bool SAMECUT, bool BRANCHY, bool HALF = false, bool ALCH = false, bool NANMASK = false>
__device__ __forceinline__ void pair_eval(float xi, float yi, float zi, float qi, float sgi, float epi,
                                          const float4 pj, const float2 lj, bool allowed, const NbParams& p,
                                          float& fx, float& fy, float& fz, float& elj, float& ecoul,
                                          float* g = nullptr, float* evir = nullptr, float* ecross = nullptr,
                                          float r2bias = 0.f, float* r2_out = nullptr, float* edudl = nullptr,
                                          const float4* __restrict__ etab = nullptr) {
    static_assert(!NANMASK || BRANCHY, "the NaN-coded exclusion needs the early-out");
    const float dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;   // tgt - src (src/cuda/util.cu:118-140)
    const float r2 = NANMASK ? __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, r2bias)))
                             : dx * dx + dy * dy + dz * dz;
    if (NANMASK) allowed = true;
    if (r2_out) *r2_out = r2;
    const bool in_lj = (r2 < p.rc2_lj) && allowed;
    const bool in_c = SAMECUT ? in_lj : ((r2 < p.rc2_coul) && allowed);
    if (BRANCHY && !(in_lj || in_c)) return;
    const float sig = GEOM ? sgi * lj.x : sgi + lj.x;
    const float eps_s = epi * lj.y;                 // 24 eps_ij (ALCH: signed)
    const float eps = ALCH ? fabsf(eps_s) : eps_s;
    // ALCH: a pair with exactly one atom in the coupled molecule.  Soft core (Beutler et al. 1994): it interacts at
    // r_sc = (alpha lambda sigma^6 + r^6)^(1/6) instead of r - LJ and Coulomb alike - scaled by 1 - lambda, so that
    // dU/dlambda stays finite at lambda = 1 (the reference's last window, src/properties/water_sol.rs:52-56).
    const bool cross = ALCH && __float_as_int(eps_s) < 0;
    const float ascale = cross ? p.alch_scale : 1.0f;
    float rinv, r2e = r2, sg6 = 0.f;
    // Ewald force table (force-only, non-alchemical flavours): its LDS read goes out FIRST - index and fraction need nothing but
    // r^2 - so the rsq and the Lennard-Jones arithmetic below run under its latency (with the lookup in source order the compiler
    // parked an s_waitcnt two instructions behind the read).  `etab` arrives biased by the first binade's offset (nb_cluster_kernel),
    // and the masked bits serve both as the entry's byte offset and as the entry's left edge: 5 VALU + the 3 FMAs below.
    float4 en = make_float4(0.f, 0.f, 0.f, 0.f); float dxt = 0.f;
    constexpr bool tab = COUL == CM_EWALD_TAB && !ENERGY && !ALCH;      // (compile time: one straight block for the scheduler)
    if (tab) {
        const float x = __builtin_fmaf(r2, p.etab_scale, EWALD_TAB_C);
        const uint32_t sh = p.etab_shift, xm = __float_as_uint(x) & (0xFFFFFFFFu << sh);
        en = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(etab) + (xm >> (sh - 4u)));
        dxt = x - __uint_as_float(xm);
    }
    if (ALCH) {
        const float sgsc = (sig > 0.f && eps != 0.f) ? sig : p.sc_sigmin;
        const float sg2 = sgsc * sgsc;
        sg6 = sg2 * sg2 * sg2;
        const float rsc6 = __builtin_fmaf(cross ? p.sc_al : 0.f, sg6, r2 * r2 * r2);
        rinv = (cross && p.sc_al != 0.f) ? __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(rsc6) * (-1.0f / 6.0f)) : __builtin_amdgcn_rsqf(r2);
        if (cross && p.sc_al != 0.f) r2e = __builtin_amdgcn_rcpf(rinv * rinv);
    } else rinv = __builtin_amdgcn_rsqf(r2);
    const float rinv2 = rinv * rinv;
    const float s2 = sig * sig * rinv2;
    const float s6 = s2 * s2 * s2;
    const float es6 = eps * s6;
#ifdef NB_EXP_NOLJ
    const float flj_r2 = 0.f;
#else
    const float flj_r2 = es6 * (2.0f * s6 - 1.0f);  // 24 eps (2 s12 - s6)      [force * r^2]
#endif
    const float qq = qi * pj.w;                     // k_e q_i q_j
    float fc_r2;                                    // Coulomb force * r^2
    if (COUL == CM_SHIFTED) fc_r2 = qq * rinv;
    else if (COUL == CM_SOFT) fc_r2 = qq * rinv * r2e * __frcp_rn(r2e + p.soft2);
    else if (COUL == CM_RF) fc_r2 = qq * (rinv - p.k_rf2 * r2e);
    float erfc_ar = 0.f;
    if (tab) {
        const float gsm = __builtin_fmaf(dxt, __builtin_fmaf(dxt, en.z, en.y), en.x);
        fc_r2 = qq * __builtin_fmaf(-gsm, r2e, rinv);
    } else if (COUL == CM_EWALD || COUL == CM_EWALD_TAB) {
        // erfc by Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7) sharing the exponential the force
        // needs anyway: ~10 VALU ops instead of the ~45 of libm's erfcf
        const float ar = p.alpha * r2e * rinv;
        const float ex = __expf(-ar * ar);
        const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ar);
        erfc_ar = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f)))) * ex;
        fc_r2 = qq * (erfc_ar * rinv + 1.1283791671f * p.alpha * ex);
    }
    float fs;
    if (SAMECUT) fs = (BRANCHY || in_lj) ? (flj_r2 + fc_r2) * rinv2 : 0.0f;
    else fs = (in_lj || in_c) ? ((in_lj ? flj_r2 : 0.0f) + (in_c ? fc_r2 : 0.0f)) * rinv2 : 0.0f;
    const float fsc = fs;                           // ALCH: -U'(r_sc) / r_sc of the unscaled pair
    if (ALCH) {
        const float w = r2 * rinv2;                 // (r / r_sc)^2: dr_sc/dr = (r / r_sc)^5
        fs *= cross ? ascale * w * w : 1.0f;
    }
    fx += fs * dx; fy += fs * dy; fz += fs * dz;
    if (HALF) { g[0] += fs * dx; g[1] += fs * dy; g[2] += fs * dz; }   // minus the force on j
    if (ENERGY) {
        const float e_l = es6 * (s6 - 1.0f) * (1.0f / 6.0f);  // 4 eps (s12 - s6)
        float e_c;
        if (COUL == CM_SHIFTED || COUL == CM_SOFT) e_c = qq * (rinv - p.coul_shift);
        else if (COUL == CM_RF) e_c = qq * (rinv + p.k_rf * r2e - p.coul_shift);
        else e_c = qq * erfc_ar * rinv;
        // fp32 partial sums: the callers fold them into fp64 once per chunk of 64 j-atoms
        const float u_l = in_lj ? e_l : 0.f, u_c = in_c ? e_c : 0.f;
        elj += ALCH ? ascale * u_l : u_l;
        ecoul += ALCH ? ascale * u_c : u_c;
        if (ALCH && ecross && cross) {
            *ecross += u_l + u_c;
            // dU/dlambda = -U_cross(r_sc) + (1 - lambda) U'(r_sc) alpha sigma^6 / (6 r_sc^5)
            if (edudl) *edudl += -(u_l + u_c) - ascale * fsc * p.sc_alpha * sg6 * rinv2 * rinv2 * (1.0f / 6.0f);
        }
        if (evir) *evir += fs * r2;   // r_ij . F_ij of the pair (fs = 0 outside the cutoffs)
    }
}
