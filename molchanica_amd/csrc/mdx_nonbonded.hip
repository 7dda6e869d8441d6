// mdx_nonbonded.hip — the Lennard-Jones + Coulomb pair loop (the dominant kernel of the path).
//
// Replaces the pair-force part of `MdState::step` in the absent `dynamics` crate; the formulas
// follow the only in-tree statement of them, /root/reference src/cuda/util.cu:53-63 (Coulomb,
// dir = tgt - src) and :92-140 (LJ 12-6: F = dir*24 eps (2 s^12 - s^6)/r, E = 4 eps (s^12 - s^6)).
// The reference's own (unused) kernels are one target per thread with a serial loop over ALL
// sources (src/cuda/cuda.cu:10-37, 73-102); this kernel keeps "one atom per lane" and replaces the
// O(N) inner loop by the tile pair list, with the neighbour atoms staged through LDS.
//
// Three kernels share the tile pair list (mdx_grid.hip) and the LDS staging of the neighbour atoms:
//   nb_cluster_kernel<.., HALF = true>   the default (nb_variant 5): half list, a lane is the pair (i-atom ii of
//       every i-cluster, j-atom jj of the entry), 4 or 8 waves per tile; the force on the i-atoms accumulates in
//       registers, the reaction on the 8 j-atoms of an entry leaves as ONE 24-lane f32 atomic (DPP reduction over
//       the i-lanes).  Fastest; the last bits of a force depend on the order the atomics land in.
//   nb_cluster_kernel<.., HALF = false>  nb_variant 2-4: same lane mapping over a full list - every pair is seen
//       from both sides, one owner per force component, no atomics, bitwise reproducible.
//   nb_tile_kernel                       nb_variant 1: lane = i-atom, j broadcast from LDS, branch-free; the A/B
//       baseline the cluster kernels are measured against.
// The step loop runs the half-list kernel over a DUAL pair list (DUAL template argument): most
// steps walk the inner list (cluster pairs with an atom pair within cutoff + inner_skin at the last pruning pass), and
// when the drift pass reports a path length above inner_skin/2 the same kernel walks the Verlet list and rebuilds the
// inner list on the side.  Energies, the minimiser and single points always walk the Verlet list.
// Common to all (gfx950, wave64):
//   * the tile's list is consumed in chunks of 8 entries = 64 j-atoms: lane l fetches j-atom l of the chunk
//     (8 x 128-B contiguous cluster records of posq + 8 x 64 B of lj: coalesced), adds the periodic image shift
//     once, and parks it in the wave's private LDS strip (no workgroup barrier in the loop); entries of chunk c+2,
//     atoms and exclusion masks of chunk c+1 are in flight while chunk c is evaluated.
//   * 29 VALU ops per in-range pair behind an exec-masked early-out (8 ops for a cluster pair with no lane inside
//     the cutoff); exclusions/self pairs only in the first n_masked entries of a tile, which carry per-lane masks
//     (an excluded lane's bit enters r^2 as a NaN addend and fails the cutoff test by itself).
//   * blockIdx is remapped so that each XCD walks one contiguous eighth of the tile range and its private L2 sees
//     a compact spatial region.
// No MFMA: this is pairwise scalar work.  Roofline: fp32 VALU bound (with the PMC numbers).
#include "mdx_internal.h"
#include "mdx_bonded_dev.h"
#include "mdx_pair_dev.h"
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <vector>

// the kernels and their launcher live in mdx_nonbonded_impl.h and are instantiated per (ENERGY, COUL) in mdx_nb_inst.hip
template <bool ENERGY, int COUL>
void launch_variant(mdx_handle* h, const NbArgs& a, bool geom, bool samecut);
#define NB_EXTERN(E, C) extern template void launch_variant<E, C>(mdx_handle*, const NbArgs&, bool, bool);
NB_EXTERN(false, CM_SHIFTED) NB_EXTERN(false, CM_RF) NB_EXTERN(false, CM_EWALD) NB_EXTERN(false, CM_SOFT)
NB_EXTERN(false, CM_EWALD_TAB)
NB_EXTERN(true, CM_SHIFTED) NB_EXTERN(true, CM_RF) NB_EXTERN(true, CM_EWALD) NB_EXTERN(true, CM_SOFT)
#undef NB_EXTERN

static bool cut_on(float rc) { return rc > 0.f && std::isfinite(rc); }
#define FAIL_NB(msg) do { mdx_set_error(msg); return MDX_EDEVICE; } while (0)

// cutoffs, image shifts and the Coulomb treatment of a handle as the pair kernels take them (also mdx_groups.hip)
void mdx_fill_nb_params(const mdx_handle* h, NbParams& p, int* mode_out, bool* geom_out, bool* samecut_out) {
    const mdx_config& c = h->cfg;
    p.rc2_lj = cut_on(c.lj_cutoff) ? c.lj_cutoff * c.lj_cutoff : FLT_MAX;
    p.rc2_coul = cut_on(c.coulomb_cutoff) ? c.coulomb_cutoff * c.coulomb_cutoff : FLT_MAX;
    for (int d = 0; d < 3; ++d) p.shift[d] = h->per[d] ? (h->box_hi[d] - h->box_lo[d]) : 0.f;
    const bool ccut = cut_on(c.coulomb_cutoff);
    const float rc = c.coulomb_cutoff;
    p.alpha = c.ewald_alpha; p.soft2 = c.softening_sq;
    p.k_rf = 0.f; p.k_rf2 = 0.f; p.coul_shift = 0.f;
    p.alch_scale = h->alch_on ? (float)(1.0 - h->alch_lambda) : 1.0f;
    p.sc_alpha = h->alch_on ? h->sc_alpha : 0.f; p.sc_al = h->alch_on ? (float)(h->sc_alpha * h->alch_lambda) : 0.f;
    p.sc_sigmin = h->sc_sigma_min;
    int mode = CM_SHIFTED;
    switch (c.coulomb_mode) {
    case MDX_COULOMB_REACTION:
        mode = CM_RF;
        if (ccut) { p.k_rf = 1.0f / (2.0f * rc * rc * rc); p.k_rf2 = 2.0f * p.k_rf; p.coul_shift = 1.5f / rc; }
        break;
    case MDX_COULOMB_EWALD: mode = CM_EWALD; break;
    default:
        mode = (c.softening_sq != 0.f) ? CM_SOFT : CM_SHIFTED;
        if (ccut) p.coul_shift = 1.0f / rc;
    }
    *mode_out = mode; *geom_out = c.combining_rule == MDX_COMBINE_GEOMETRIC; *samecut_out = p.rc2_lj == p.rc2_coul;
    // Ewald real space: the table of the smooth part, where the handle has one (mdx_build_ewald_table)
    p.etab = mode == CM_EWALD ? h->d.ewald_tab : nullptr;
    p.etab_n = p.etab ? h->ewald_tab_n : 0u;
    p.etab_scale = h->ewald_tab_scale; p.etab_shift = h->ewald_tab_shift;
}

// g(r^2) = [erf(beta r)/r - 2 beta/sqrt(pi) exp(-beta^2 r^2)] / r^2 over the intervals of the bit-indexed table (mdx_pair_dev.h), in fp64
int mdx_build_ewald_table(mdx_handle* h) {
    if (h->d.ewald_tab) { (void)hipFree(h->d.ewald_tab); h->d.ewald_tab = nullptr; }
    h->ewald_tab_n = 0;
    const mdx_config& c = h->cfg;
    if (c.coulomb_mode != MDX_COULOMB_EWALD || !cut_on(c.lj_cutoff) || !cut_on(c.coulomb_cutoff)) return MDX_OK;
    if (const char* e = std::getenv("MDX_EWALD_TABLE")) { if (e[0] == '0') return MDX_OK; }      // the closed form (A/B; read per handle, at create)
    const double beta = c.ewald_alpha, rmax = std::max(c.lj_cutoff, c.coulomb_cutoff);
    const double scale = std::max(1.0, (beta / 0.3) * (beta / 0.3));
    const uint32_t sh = beta > 0.36 ? 16u : 17u;      // mantissa bits kept: 7 or 6
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto flt = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    const uint32_t off = EWALD_TAB_CBITS >> sh;
    const uint32_t k_last = (bits((float)(rmax * rmax * 1.0001 * scale) + EWALD_TAB_C) >> sh) - off;
    if (k_last + 2 > EWALD_TAB_MAX) return MDX_OK;      // beta rc far beyond the usual 3: the closed form stays
    auto g = [&](double x) {
        const double r2 = std::max(0.0, (x - (double)EWALD_TAB_C) / scale), z2 = beta * beta * r2;
        if (z2 < 1e-2) return beta * beta * beta * 1.1283791670955126 * (2.0 / 3.0 - 0.4 * z2 + z2 * z2 / 7.0 - z2 * z2 * z2 / 27.0);
        const double r = std::sqrt(r2);
        return (std::erf(beta * r) / r - 1.1283791670955126 * beta * std::exp(-z2)) / r2;
    };
    std::vector<float4> tab(k_last + 2);
    for (uint32_t k = 0; k < tab.size(); ++k) {      // the parabola through both ends and the middle of the interval
        const double x0 = flt((k + off) << sh), x1 = flt((k + 1 + off) << sh), hh = x1 - x0;
        const double g0 = g(x0), gm = g(x0 + 0.5 * hh), g1 = g(x1);
        const double cq = 2.0 * (g0 - 2.0 * gm + g1) / (hh * hh), bq = (g1 - g0) / hh - cq * hh;
        tab[k] = make_float4((float)g0, (float)bq, (float)cq, 0.f);
    }
    HIP_TRY(hipMalloc((void**)&h->d.ewald_tab, sizeof(float4) * tab.size()));
    HIP_TRY(hipMemcpy(h->d.ewald_tab, tab.data(), sizeof(float4) * tab.size(), hipMemcpyHostToDevice));
    h->ewald_tab_n = (uint32_t)tab.size(); h->ewald_tab_scale = (float)scale; h->ewald_tab_shift = sh;
    return MDX_OK;
}

// Single-device handles with velocity Verlet and no constraints / virtual sites / SPME / external forces / alchemy, dual list with split
// path arrays.  By default the eight-waves-per-tile class (below 2048 tiles: a step there is launch latency, and this is one launch
// instead of two); MDX_ONEPASS=2 also takes the one-wave class (1 M atoms: measured slower - the pair launch grows by more than the
// 35 us pass it absorbs); MDX_ONEPASS=0: off.  Read per chunk.
bool mdx_onepass_ok(const mdx_handle* h) {
    const char* e = std::getenv("MDX_ONEPASS");
    if (e && e[0] == '0') return false;
    const int wpt = mdx_nb_wpt_half(h, h->T);
    if (!(wpt == 8 || (wpt == 1 && e && e[0] == '2'))) return false;
    static const bool dbg = [] { const char* x = std::getenv("MDX_DEBUG_ONEPASS"); return x && x[0] == '1'; }();
    if (dbg) std::fprintf(stderr, "[mdx onepass] refused %d dd %d dual %d path_split %d force_b %d fuse_ok %d variant %d wpt %d (T %u)\n", (int)h->onepass_refused, h->dd != nullptr,
                          (int)h->dual_on, (int)h->path_split, h->d.force_b != nullptr, (int)mdx_bonded_integrate_ok(h), mdx_nb_variant(h), mdx_nb_wpt_half(h, h->T), h->T);
    if (h->onepass_refused || h->dd || !h->dual_on || !h->path_split || !h->d.force_b) return false;
    if (mdx_nb_variant(h) != 5 || h->integrator != MDX_INTEGRATOR_VERLET_VELOCITY || mdx_has_constraints(h) || h->n_vsites != 0 || h->pme_on ||
        h->have_ext || h->n_local != h->N || h->alch_on || h->d.posq_alt == nullptr) return false;
    if (wpt == 1 && !mdx_bonded_integrate_ok(h)) return false;
    if (h->profile && h->profile_level < 2) return false;      // (event brackets around every kernel count launches per kind: they keep the separate passes)
    // the pair kernel's flavours that exist in this form (mdx_nonbonded_impl.h NB_STEP): orthorhombic periodic cells, not the softened Coulomb
    NbParams p; int mode = CM_SHIFTED; bool geom = false, samecut = false;
    mdx_fill_nb_params(h, p, &mode, &geom, &samecut);
    return !geom && mode != CM_SOFT;
}

int mdx_launch_nonbonded(mdx_handle* h, bool energy, const uint32_t* d_gate, uint32_t thr_bits, int part) {
    const mdx_config& c = h->cfg;
    NbArgs a{};
    a.T = h->T;
    if (part != 0) {
        if (!h->tile_split) FAIL_NB("internal: split pair-kernel launch without a tile classification");
        a.tile_order = h->d.tile_order;
        a.t_first = part == 1 ? 0u : h->n_interior;
        a.t_count = part == 1 ? h->n_interior : h->T - h->n_interior;
    }
    if (part == 0 && h->tile_lpt_on) { a.tile_order = h->d.tile_lpt; a.t_first = 0; a.t_count = h->T; }   // longest lists first
    a.posq = h->d.posq; a.lj = h->d.lj; a.counts = h->d.list_counts; a.entry_off = h->d.entry_off;
    a.mchunk_off = h->d.mchunk_off; a.entries = h->d.entries; a.masks = h->d.masks;
    a.force = h->d.force; a.energy = h->d.energy; a.slot_flags = h->d.slot_flags; a.gate = d_gate; a.thr_bits = thr_bits;
    a.energy_all = mdx_dd_half_shell(h) ? 1u : 0u;
    {
        static const int xcd_env = [] { const char* e = std::getenv("MDX_XCD_INTERLEAVE"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
        a.xcd_interleave = xcd_env >= 0 ? (uint32_t)xcd_env : ((h->have_local_bounds && h->n_local != h->N) ? 1u : 0u);
        if (part == 0 && h->tile_lpt_on && !h->tile_lpt_grouped) a.xcd_interleave = 1u;    // (an order by length has no spatial runs to keep on one XCD)
    }
    // dual list: only force calls of the step loop (nb_step >= 0) use the inner masks; everything else - energies, the
    // minimiser, the first evaluation after a rebuild - walks the plain list, which is always valid
    // ... except the one that finishes the step behind a rebuild (nb_post_rebuild): that one IS the pruning pass (round 3: it
    // used to walk the plain list, 514 us at 1 M atoms, and the next step pruned, 604; now 604 here and 445 there)
    const bool step_call = h->nb_step >= 0 && d_gate != nullptr;
    // (round 6: ... or, where the rebuild's own pruning pass wrote the inner list - one wave per tile, single device - an inner-list walk)
    a.inner = (h->dual_on && !h->alch_on && !energy && (step_call || (h->nb_post_rebuild && (h->prune_pending || h->inner_from_rebuild)))) ? 1u : 0u;
    if (a.inner) {
        if (part != 2) { h->prune_latch = h->prune_pending; h->prune_pending = false; }   // one decision for both halves of a split launch
        a.force_prune = h->prune_latch ? 1u : 0u;
        a.prune_flag = &h->d.ctl->prune[step_call ? h->nb_step + 1 : 0];   // (word 0 is never raised: the drift of step s writes s + 1)
        a.prune_flag2 = (part == 2 && step_call) ? &h->d.ctl->prune_ghost[h->nb_step + 1] : nullptr;
        const float rin = std::max(cut_on(c.lj_cutoff) ? c.lj_cutoff : 0.f, cut_on(c.coulomb_cutoff) ? c.coulomb_cutoff : 0.f) + h->inner_skin;
        a.rin2 = rin * rin;
        a.entries_in = h->d.entries_in; a.inner_nch = h->d.inner_nch; a.ref = h->d.ref; a.inner_count = h->d.inner_count;
        a.path = h->path_split ? h->d.path : nullptr; a.dprune = h->path_split ? h->d.dprune : nullptr;
        // (a launch gated off behind a stale list is followed by a rebuild, which sets prune_pending again)
    }
    // small systems: the bonded gather rides along as extra workgroups of the merged dual-list launch (launch_variant
    // decides; mdx_launch_bonded then finds bonded_fused set).  MDX_FUSE_BONDED=0: A/B knob.
    static const bool fuse_bonded = [] { const char* e = std::getenv("MDX_FUSE_BONDED"); return !(e && e[0] == '0'); }();
    h->bonded_fused = false;
    if (fuse_bonded && a.inner && part == 0 && mdx_bonded_wanted(h) && !h->bonded_deferred && (!h->profile || h->profile_level >= 2)) {
        a.b_S = h->S; a.b_role_off = h->d.role_off_s; a.b_roles = h->d.role_rec_s; a.b_prm = h->d.role_prm;
        mdx_fill_bonded_params(h, a.b_p);
    }
    if (h->onepass.fprev && a.inner && part == 0 && !energy) {      // one launch per step (mdx_step "onepass")
        const OnePassNow& o = h->onepass;
        a.posq = o.yin; a.force = o.fcur;
        a.st_fprev = o.fprev; a.st_yout = o.yout; a.st_fnext = o.fnext; a.st_vel = h->d.vel; a.st_dt = o.dt; a.st_last = o.last ? 1u : 0u; a.st_kick = o.first ? 0.5f : 1.0f;
        {   // MDX_ONEPASS_GRANT=x (tests): scales what the words grant the unknown kick; negative: the words under-estimate, launches contradict them
            static const float grant = [] { const char* e = std::getenv("MDX_ONEPASS_GRANT"); return e ? (float)std::atof(e) : 1.0f; }();
            a.st_grant = grant;
        }
        a.st_disp_out = o.disp_out; a.st_prune_out = o.prune_out; a.st_viol = o.viol;
        a.st_path_thr = 0.5f * h->inner_skin * (1.0f - 1.0e-4f);
        a.b_S = 0; a.b_role_off = nullptr;
        if (mdx_bonded_wanted(h)) {      // one wave per tile: the tile's wave evaluates its atoms' roles itself; eight: extra workgroups (b_S)
            a.b_role_off = h->d.role_off_s; a.b_roles = h->d.role_rec_s; a.b_prm = h->d.role_prm;
            mdx_fill_bonded_params(h, a.b_p);
            if (mdx_nb_wpt_half(h, a.T) != 1) a.b_S = h->S;
        }
        h->force_zeroed = true;          // (the launch before this one - or the chunk's opening pass - zeroed the buffer this one accumulates into)
        ++h->onepass_launches;
    }
    int mode = CM_SHIFTED; bool geom = false, samecut = false;
    mdx_fill_nb_params(h, a.p, &mode, &geom, &samecut);
    mdx_prof_begin(h, energy ? 3 : (part == 2 ? 4 : 0));   // 3: the energy flavour is a different kernel, keep it out of the step-loop average
    // the half-list kernel accumulates with atomics: start from zero (part of the kernel's cost, so
    // inside the profiled bracket; harmless when the launch behind it is gated off, see mdx_step)
    if (part != 2) {
        if (mdx_nb_half(h) && !h->force_zeroed) HIP_TRY(hipMemsetAsync(h->d.force, 0, sizeof(float4) * (size_t)h->S, h->stream));
        h->force_zeroed = false;
    }
#define NB_DISPATCH(E)                                                              \
    switch (mode) {                                                                 \
    case CM_SHIFTED: launch_variant<E, CM_SHIFTED>(h, a, geom, samecut); break;     \
    case CM_SOFT: launch_variant<E, CM_SOFT>(h, a, geom, samecut); break;           \
    case CM_RF: launch_variant<E, CM_RF>(h, a, geom, samecut); break;               \
    default: launch_variant<E, CM_EWALD>(h, a, geom, samecut); break;               \
    }
    // (force-only Ewald launches of the cluster kernels read the smooth part from the table: a flavour of its own, so that the
    // lookup is straight-line code - as a run-time branch inside CM_EWALD it cost two scalar branches per cluster pair and pinned
    // the LDS read's wait right behind the read)
    if (!energy && mode == CM_EWALD && a.p.etab && !h->alch_on && mdx_nb_variant(h) != 1) launch_variant<false, CM_EWALD_TAB>(h, a, geom, samecut);
    else if (energy) { NB_DISPATCH(true) } else { NB_DISPATCH(false) }
#undef NB_DISPATCH
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}
