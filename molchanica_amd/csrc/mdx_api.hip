// mdx_api.hip — handle life-cycle and the C ABI of include/mdx.h.
//
// Mirrors the surface of the (absent) `dynamics` crate that Molchanica consumes:
//   MdState::new -> mdx_create            [ref: /root/reference src/md/mod.rs:689]
//   MdState::step -> mdx_step             [ref: src/md/mod.rs:716,748; src/mol_alignment.rs:346]
//   compute_energy_snapshot -> mdx_single_point / mdx_energy   [ref: src/md/mod.rs:1036]
//   md.atoms[i].posit/.force -> mdx_download/mdx_upload        [ref: src/mol_alignment.rs:349-352]
//   md.cell + rebuild_spatial_caches -> mdx_set_box / mdx_rebuild_spatial_caches
//                                                   [ref: src/properties/sol_shrinking_box.rs:600-632]
// There is no CPU fallback in this library: without a gfx950 device every entry point fails with
// MDX_EDEVICE and the host keeps its own CPU path (src/util.rs:1072-1119 semantics).
#include "mdx_comm.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <unordered_map>

static thread_local std::string g_last_error;
void mdx_set_error(const std::string& s) { g_last_error = s; }

static inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }
static bool cut_on(float rc) { return rc > 0.f && std::isfinite(rc); }
static uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

#define FAIL(code, msg) do { mdx_set_error(msg); return (code); } while (0)

template <typename T>
static int upload_vec(T** dptr, const std::vector<T>& v, hipStream_t st) {
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    HIP_TRY(hipMalloc((void**)dptr, std::max<size_t>(sizeof(T) * v.size(), 16)));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(*dptr, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, st));
    return MDX_OK;
}
template <typename T>
static int alloc_n(T** dptr, size_t n) {
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    HIP_TRY(hipMalloc((void**)dptr, std::max<size_t>(sizeof(T) * n, 16)));
    return MDX_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int mdx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* mdx_last_error(void) { return g_last_error.c_str(); }

extern "C" void mdx_config_default(mdx_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->lj_cutoff = 10.0f; c->coulomb_cutoff = 10.0f; c->skin = 2.0f;
    c->coulomb_k = 332.0637f; c->scale14_lj = 0.5f; c->scale14_coulomb = 1.0f / 1.2f;
    c->coulomb_mode = MDX_COULOMB_SHIFTED; c->combining_rule = MDX_COMBINE_LORENTZ_BERTHELOT;
    c->overrides = MDX_OVR_LONG_RANGE_RECIP_DISABLED;
    c->chunk_steps = 16;
    c->constraint_tol = 1e-5f; c->constraint_max_iter = 64;
}

// ---------------------------------------------------------------------------------------------
static int validate(const mdx_system* s, const mdx_config* c) {
    if (!s || !c) FAIL(MDX_EPARAM, "null system or config");
    const uint32_t N = s->n_atoms;
    if (N == 0) FAIL(MDX_EPARAM, "system has no atoms");
    if (N > (1u << 30)) FAIL(MDX_EPARAM, "too many atoms");
    if (!s->pos || !s->mass || !s->charge || !s->lj_type || !s->lj_sigma || !s->lj_eps)
        FAIL(MDX_EPARAM, "missing per-atom array (pos/mass/charge/lj_type/lj_sigma/lj_eps)");
    if (s->n_lj_types == 0) FAIL(MDX_EPARAM, "no LJ types");
    for (uint32_t i = 0; i < N; ++i) {
        if (s->lj_type[i] >= s->n_lj_types) FAIL(MDX_EPARAM, "lj_type index out of range");
        const bool fixed = s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
        if (!fixed && !(s->mass[i] > 0.f && std::isfinite(s->mass[i]))) FAIL(MDX_EPARAM, "non-positive mass");
        if (!std::isfinite(s->charge[i])) FAIL(MDX_EPARAM, "non-finite charge");
        for (int d = 0; d < 3; ++d)
            if (!std::isfinite(s->pos[3 * i + d])) FAIL(MDX_EPARAM, "non-finite position");
    }
    for (uint32_t t = 0; t < s->n_lj_types; ++t)
        if (!(s->lj_sigma[t] >= 0.f) || !(s->lj_eps[t] >= 0.f) || !std::isfinite(s->lj_sigma[t]) ||
            !std::isfinite(s->lj_eps[t]))
            FAIL(MDX_EPARAM, "LJ sigma/eps must be finite and non-negative");
    auto chk = [&](const uint32_t* idx, uint32_t n, int w, const char* what) -> int {
        if (n && !idx) FAIL(MDX_EPARAM, std::string("missing index array: ") + what);
        for (size_t k = 0; k < (size_t)n * w; ++k)
            if (idx[k] >= N) FAIL(MDX_EPARAM, std::string("atom index out of range in ") + what);
        return MDX_OK;
    };
    MDX_TRY(chk(s->bond_idx, s->n_bonds, 2, "bonds"));
    MDX_TRY(chk(s->angle_idx, s->n_angles, 3, "angles"));
    MDX_TRY(chk(s->dihedral_idx, s->n_dihedrals, 4, "dihedrals"));
    MDX_TRY(chk(s->pairs14_idx, s->n_pairs14, 2, "pairs14"));
    if (s->n_bonds && (!s->bond_k || !s->bond_r0)) FAIL(MDX_EPARAM, "missing bond parameters");
    if (s->n_angles && (!s->angle_k || !s->angle_theta0)) FAIL(MDX_EPARAM, "missing angle parameters");
    if (s->n_dihedrals && (!s->dihedral_v || !s->dihedral_phase || !s->dihedral_n))
        FAIL(MDX_EPARAM, "missing dihedral parameters");
    if (s->excl_offsets) {
        if (s->excl_offsets[0] != 0) FAIL(MDX_EPARAM, "excl_offsets[0] must be 0");
        for (uint32_t i = 0; i < N; ++i)
            if (s->excl_offsets[i + 1] < s->excl_offsets[i]) FAIL(MDX_EPARAM, "excl_offsets not monotone");
        if (s->excl_offsets[N] && !s->excl_idx) FAIL(MDX_EPARAM, "missing excl_idx");
        for (uint32_t k = 0; k < s->excl_offsets[N]; ++k)
            if (s->excl_idx[k] >= N) FAIL(MDX_EPARAM, "exclusion index out of range");
    }
    if (!(c->skin >= 0.f) || !std::isfinite(c->skin)) FAIL(MDX_EPARAM, "skin must be finite and >= 0");
    if (!(c->coulomb_k >= 0.f)) FAIL(MDX_EPARAM, "coulomb_k must be >= 0");
    if (c->coulomb_mode < 0 || c->coulomb_mode > 2) FAIL(MDX_EPARAM, "unknown coulomb_mode");
    if (c->combining_rule < 0 || c->combining_rule > 1) FAIL(MDX_EPARAM, "unknown combining_rule");
    if (c->coulomb_mode == MDX_COULOMB_EWALD && !(c->ewald_alpha > 0.f))
        FAIL(MDX_EPARAM, "ewald_alpha must be > 0 for MDX_COULOMB_EWALD");
    return MDX_OK;
}

static void decode_periodic(int32_t v, int per[3]) {
    if (v & 0x10) { per[0] = v & 1; per[1] = (v >> 1) & 1; per[2] = (v >> 2) & 1; }
    else per[0] = per[1] = per[2] = (v != 0);
}

static int check_box(const int per[3], const float* lo, const float* hi, const mdx_config* c) {
    if (!(per[0] || per[1] || per[2])) return MDX_OK;
    if (!cut_on(c->lj_cutoff) || !cut_on(c->coulomb_cutoff))
        FAIL(MDX_EPARAM, "a periodic system needs finite LJ and Coulomb cut-offs");
    const float rl = std::max(c->lj_cutoff, c->coulomb_cutoff) + c->skin;
    for (int d = 0; d < 3; ++d) {
        if (!per[d]) continue;
        const float L = hi[d] - lo[d];
        if (!(L > 0.f) || !std::isfinite(L)) FAIL(MDX_EPARAM, "box extent must be positive");
        if (L < 2.0f * rl)
            FAIL(MDX_EPARAM, "box edge shorter than 2*(cutoff+skin): minimum image is not unique");
    }
    return MDX_OK;
}

int mdx_check_box(const mdx_handle* h, const float* lo, const float* hi) { return check_box(h->per, lo, hi, &h->cfg); }

static void free_device(mdx_handle* h) {
    DeviceState& d = h->d;
    void* ptrs[] = {d.o_qs, d.o_lj, d.o_invm, d.o_mass, d.o_q, d.o_lj_raw, d.excl_off, d.excl_idx, d.pos_orig,
                    d.vel_orig, d.ext_orig, d.posq, d.posq_alt, d.lj, d.vel, d.force, d.ref, d.orig_of, d.slot_of, d.gid, d.lflag,
                    d.slot_flags, d.cell_of,
                    d.cell_count, d.cell_start, d.cell_cursor, d.sorted_orig, d.sorted_tmp, d.col_tiles, d.tile_start,
                    d.tile_col, d.scan_tmp, d.cl_lo, d.cl_hi, d.cl_kind, d.list_counts, d.entry_cnt, d.entry_off,
                    d.mchunk_cnt, d.mchunk_off, d.entries, d.entries_in, d.inner_nch, d.list_cursors, d.masks, d.role_off_o, d.role_rec_o, d.role_cnt_s,
                    d.role_off_s, d.role_rec_s, d.role_prm, d.ctl, d.energy,
                    d.flags_dev, d.bbox_red, d.pair_count, d.inner_count, d.pme_force, d.wstep_s, d.path, d.dprune, d.force_b, d.force_c, d.cons_o, d.cons_s, d.cons_tmp, d.cons_mask, d.cons_cnt, d.cons_off, d.cons_vir, d.vsite_o, d.vsite_s, d.gsite_o, d.gsite_s, d.gsite_tmp, d.pme_q, d.pme_f,
                    d.pme_theta, d.pme_q2, d.pme_f2, d.scratch4, d.tile_bnd, d.tile_scan, d.tile_order, d.tile_lpt, d.rb_ctl, d.scan_chain, d.grp, d.grp_mat, d.star_o, d.star_s, d.ewald_tab};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d = DeviceState{};
}

extern "C" void mdx_destroy(mdx_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    mdx_dd_destroy(h);
    for (auto& e : h->ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : h->ev_pool) (void)hipEventDestroy(e);
    mdx_pme_destroy(h);
    free_device(h);
    if (h->h_ctl) (void)hipHostFree(h->h_ctl);
    if (h->h_rb) (void)hipHostFree(h->h_rb);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int create_impl(const mdx_system* s, const mdx_config* c, int device, mdx_handle* h) {
    const uint32_t N = s->n_atoms;
    h->N = N; h->cfg = *c; h->device = device;
    // 0 = the library chooses: a chunk costs one host synchronisation (~15 us), which is 2 % of a 50 us step at 16 steps per
    // chunk; the chunk-length prediction (mdx_step) keeps longer chunks from running far past a stale list
    if (h->cfg.chunk_steps == 0) h->cfg.chunk_steps = N < 131072u ? 48u : 16u;
    if (h->cfg.chunk_steps > MDX_MAX_CHUNK) h->cfg.chunk_steps = MDX_MAX_CHUNK;
    decode_periodic(s->periodic, h->per);
    h->periodic = h->per[0] || h->per[1] || h->per[2];
    for (int d = 0; d < 3; ++d) { h->box_lo[d] = s->box_lo[d]; h->box_hi[d] = s->box_hi[d]; }
    MDX_TRY(check_box(h->per, h->box_lo, h->box_hi, c));
    h->n_local = N; h->cap_local = N;
    const bool all_cut = cut_on(c->lj_cutoff) && cut_on(c->coulomb_cutoff);
    if (c->skin == 0.f && all_cut && h->periodic) {      // the library chooses (and tunes, mdx_step): start from the usual 2 A, or what the box allows
        float s0 = 2.0f;
        for (int d = 0; d < 3; ++d)
            if (h->per[d]) s0 = std::min(s0, 0.5f * (h->box_hi[d] - h->box_lo[d]) - std::max(c->lj_cutoff, c->coulomb_cutoff) - 0.01f);
        if (s0 >= 0.5f) { h->cfg.skin = s0; h->skin_tune.on = true; h->skin_tune.base_skin = s0; MDX_TRY(check_box(h->per, h->box_lo, h->box_hi, &h->cfg)); }
    }
    h->r_list = all_cut ? std::max(c->lj_cutoff, c->coulomb_cutoff) + h->cfg.skin : INFINITY;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) FAIL(MDX_EDEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) FAIL(MDX_EDEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    h->pme_cus_per_xcd = mdx_pme_cu_split(h, c);
    if (h->pme_cus_per_xcd) MDX_TRY(mdx_stream_create_masked(&h->stream, h->pme_cus_per_xcd, true));
    else HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    hipStream_t st = h->stream;
    DeviceState& d = h->d;

    // ---- static per-atom data ----
    const bool lj_off = (c->overrides & MDX_OVR_LJ_DISABLED) != 0;
    const bool coul_off = (c->overrides & MDX_OVR_COULOMB_DISABLED) != 0;
    const bool geom = c->combining_rule == MDX_COMBINE_GEOMETRIC;
    const float sqrt_ke = std::sqrt(c->coulomb_k);
    std::vector<float> qs(N), invm(N), mass(N), q(N);
    std::vector<float2> lj(N), ljraw(N);
    h->flags.assign(N, 0);
    h->n_mobile = 0; h->total_mass = 0.0;
    for (uint32_t i = 0; i < N; ++i) {
        const uint8_t fl = s->flags ? s->flags[i] : 0;
        h->flags[i] = fl;
        const bool nb_off = (fl & MDX_ATOM_BONDED_ONLY) != 0;
        const bool fixed = (fl & (MDX_ATOM_STATIC | MDX_ATOM_GHOST)) != 0;
        const float sg = s->lj_sigma[s->lj_type[i]], ep = s->lj_eps[s->lj_type[i]];
        q[i] = (nb_off || coul_off) ? 0.f : s->charge[i];
        qs[i] = q[i] * sqrt_ke;
        const float ep_eff = (nb_off || lj_off) ? 0.f : ep;
        lj[i] = make_float2(geom ? std::sqrt(sg) : 0.5f * sg, std::sqrt(24.0f * ep_eff));
        ljraw[i] = make_float2(sg, ep_eff);
        mass[i] = s->mass[i];
        invm[i] = fixed ? 0.f : MDX_ACC_CONV / s->mass[i];
        if (!fixed) { h->n_mobile++; }
        if (!(fl & MDX_ATOM_GHOST)) h->total_mass += s->mass[i];
    }
    h->h_mass = mass; h->h_lj = lj;
    {   // Clusters by interaction kind (mdx_grid.hip, rb_assign_kernel): worth it when the system has a sizeable share of atoms with
        // a Lennard-Jones well and no charge AND of charged atoms without a well - four-site water (OPC / TIP4P: the oxygen carries
        // the well, hydrogens and the M site the charges): 6 of the 16 site pairs of two such waters do not interact at all.
        uint32_t n_lj_only = 0, n_q_only = 0;
        for (uint32_t i = 0; i < N; ++i) {
            const bool has_lj = lj[i].y != 0.f, has_q = qs[i] != 0.f;
            n_lj_only += has_lj && !has_q; n_q_only += has_q && !has_lj;
        }
        const char* const e = std::getenv("MDX_KIND_CLUSTERS");
        h->kind_split = e ? e[0] != '0' : (n_lj_only >= N / 20u && n_q_only >= N / 20u && N >= 200000u);      // (23 k sites: 6780 steps/s without, 6650 with)
    }
    h->mol_start.assign(s->mol_start ? s->mol_start : nullptr, s->mol_start ? s->mol_start + s->n_mols : nullptr);
    h->total_charge = 0.0; h->sum_q2 = 0.0;
    h->q_abs_max = 0.0;
    for (uint32_t i = 0; i < N; ++i) { h->total_charge += q[i]; h->sum_q2 += (double)q[i] * q[i]; h->q_abs_max = std::max(h->q_abs_max, (double)std::fabs(q[i])); }
    const bool pme_on = c->coulomb_mode == MDX_COULOMB_EWALD && !(c->overrides & MDX_OVR_LONG_RANGE_RECIP_DISABLED) &&
                        !coul_off;
    MDX_TRY(upload_vec(&d.o_qs, qs, st)); MDX_TRY(upload_vec(&d.o_lj, lj, st));
    MDX_TRY(upload_vec(&d.o_invm, invm, st)); MDX_TRY(upload_vec(&d.o_mass, mass, st));
    MDX_TRY(upload_vec(&d.o_q, q, st)); MDX_TRY(upload_vec(&d.o_lj_raw, ljraw, st));

    // ---- merged exclusion CSR (1-2, 1-3 and 1-4), symmetric, sorted, unique ----
    std::vector<std::vector<uint32_t>> ex(N);
    {
        if (s->excl_offsets)
            for (uint32_t i = 0; i < N; ++i)
                for (uint32_t k = s->excl_offsets[i]; k < s->excl_offsets[i + 1]; ++k) {
                    const uint32_t j = s->excl_idx[k];
                    if (j == i) continue;
                    ex[i].push_back(j); ex[j].push_back(i);
                }
        for (uint32_t p = 0; p < s->n_pairs14; ++p) {
            const uint32_t a = s->pairs14_idx[2 * p], b = s->pairs14_idx[2 * p + 1];
            if (a == b) FAIL(MDX_EPARAM, "1-4 pair of an atom with itself");
            ex[a].push_back(b); ex[b].push_back(a);
        }
        std::vector<uint32_t> off(N + 1, 0), idx;
        for (uint32_t i = 0; i < N; ++i) {
            std::sort(ex[i].begin(), ex[i].end());
            ex[i].erase(std::unique(ex[i].begin(), ex[i].end()), ex[i].end());
            if (ex[i].size() > 255) FAIL(MDX_EPARAM, "more than 255 exclusions on one atom");
            off[i + 1] = off[i] + (uint32_t)ex[i].size();
            idx.insert(idx.end(), ex[i].begin(), ex[i].end());
        }
        MDX_TRY(upload_vec(&d.excl_off, off, st)); MDX_TRY(upload_vec(&d.excl_idx, idx, st));
    }

    // ---- bonded terms -> per-atom role lists (caller order) ----
    h->n_bonds = s->n_bonds; h->n_angles = s->n_angles; h->n_dih = s->n_dihedrals; h->n_p14 = s->n_pairs14;
    {
        std::vector<uint32_t> cnt(N + 1, 0);
        auto count_term = [&](const uint32_t* idx, uint32_t n, int w) {
            for (size_t k = 0; k < (size_t)n * w; ++k) cnt[idx[k] + 1]++;
        };
        count_term(s->bond_idx, s->n_bonds, 2); count_term(s->angle_idx, s->n_angles, 3);
        count_term(s->dihedral_idx, s->n_dihedrals, 4); count_term(s->pairs14_idx, s->n_pairs14, 2);
        // the reciprocal sum sees every pair: excluded and 1-4 partners get erf(beta r)/r removed - those that carry charge on both
        // sides (the oxygen of a four-site water has none: three of its molecule's six excluded pairs are no terms at all)
        auto recip_excluded = [&](uint32_t i, uint32_t j) { return q[i] != 0.f && q[j] != 0.f; };
        if (pme_on)
            for (uint32_t i = 0; i < N; ++i)
                for (uint32_t j : ex[i]) cnt[i + 1] += recip_excluded(i, j) ? 1u : 0u;
        for (uint32_t i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
        const uint32_t R = cnt[N];
        std::vector<RoleRec> recs(R);
        std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
        // distinct parameter sets (by bit pattern, per term kind): the records carry an index
        std::vector<float4> prm_tab;
        struct PKey { uint32_t k, a, b, c; bool operator==(const PKey& o) const { return k == o.k && a == o.a && b == o.b && c == o.c; } };
        struct PHash { size_t operator()(const PKey& x) const { uint64_t h = x.k * 0x9E3779B97F4A7C15ull; h = (h ^ x.a) * 0xBF58476D1CE4E5B9ull; h = (h ^ x.b) * 0x94D049BB133111EBull; h = (h ^ x.c) * 0x9E3779B97F4A7C15ull; return (size_t)(h ^ (h >> 29)); } };
        std::unordered_map<PKey, uint32_t, PHash> prm_index;
        bool prm_overflow = false;
        auto add_term = [&](const uint32_t* at, int w, uint32_t kind, float p0, float p1, float p2) {
            const PKey key{kind, f2u(p0), f2u(p1), f2u(p2)};
            auto it = prm_index.find(key);
            uint32_t pi;
            if (it == prm_index.end()) {
                pi = (uint32_t)prm_tab.size();
                if (pi >= (1u << 24)) { prm_overflow = true; pi = 0; }
                else { prm_index.emplace(key, pi); prm_tab.push_back(make_float4(p0, p1, p2, 0.f)); }
            } else pi = it->second;
            for (int r = 0; r < w; ++r) {
                RoleRec rec{};
                int q = 0;
                for (int k = 0; k < w; ++k) if (k != r) rec.p[q++] = at[k];
                rec.meta = kind | ((uint32_t)r << 4) | (pi << 8);
                recs[cur[at[r]]++] = rec;
            }
        };
        // an atom's roles are ordered by kind, the expensive kinds first (dihedrals, angles, bonds, 1-4 pairs): the lanes of a
        // wave walk their lists in step, and role i of neighbouring atoms is then mostly the same kind - in water (angle, bond,
        // bond) for an oxygen and (angle, bond) for a hydrogen, where (bond, bond, angle) / (bond, angle) made every wave run
        // both branches at i = 1 and i = 2
        for (uint32_t k = 0; k < s->n_dihedrals; ++k)
            add_term(s->dihedral_idx + 4 * k, 4, ROLE_DIHEDRAL, s->dihedral_v[k], s->dihedral_phase[k],
                     (float)s->dihedral_n[k]);
        for (uint32_t k = 0; k < s->n_angles; ++k)
            add_term(s->angle_idx + 3 * k, 3, ROLE_ANGLE, s->angle_k[k], s->angle_theta0[k], 0.f);
        for (uint32_t k = 0; k < s->n_bonds; ++k) {
            if (s->bond_idx[2 * k] == s->bond_idx[2 * k + 1]) FAIL(MDX_EPARAM, "bond of an atom with itself");
            add_term(s->bond_idx + 2 * k, 2, ROLE_BOND, s->bond_k[k], s->bond_r0[k], 0.f);
        }
        for (uint32_t k = 0; k < s->n_pairs14; ++k) {
            const uint32_t a = s->pairs14_idx[2 * k], b = s->pairs14_idx[2 * k + 1];
            const float sa = ljraw[a].x, sb = ljraw[b].x;
            const float sig = geom ? std::sqrt(sa * sb) : 0.5f * (sa + sb);
            const float eps = std::sqrt(ljraw[a].y * ljraw[b].y);
            add_term(s->pairs14_idx + 2 * k, 2, ROLE_PAIR14, sig, 4.0f * c->scale14_lj * eps,
                     c->scale14_coulomb * c->coulomb_k * q[a] * q[b]);
        }
        if (pme_on)
            for (uint32_t i = 0; i < N; ++i)
                for (uint32_t j : ex[i])
                    if (i < j && recip_excluded(i, j)) {
                        const uint32_t at[2] = {i, j};
                        add_term(at, 2, ROLE_EWALD_EXCL, c->coulomb_k * q[i] * q[j], 0.f, 0.f);
                    }
        h->n_roles = R;
        h->n_roles_excl = 0;
        for (const RoleRec& r : recs) h->n_roles_excl += (r.meta & 0xFu) == ROLE_EWALD_EXCL ? 1u : 0u;
        h->n_roles_dih = 0;
        for (const RoleRec& r : recs) h->n_roles_dih += ((r.meta & 0xFu) != ROLE_BOND && (r.meta & 0xFu) != ROLE_ANGLE) ? 1u : 0u;
        if (prm_overflow) FAIL(MDX_EPARAM, "more than 16.7 M distinct bonded parameter sets");
        MDX_TRY(upload_vec(&d.role_prm, prm_tab, st));
        MDX_TRY(upload_vec(&d.role_off_o, cnt, st)); MDX_TRY(upload_vec(&d.role_rec_o, recs, st));
        MDX_TRY(alloc_n(&d.role_rec_s, R));
        HIP_TRY(hipStreamSynchronize(st));
    }

    // bonded / constrained pairs, kept for the hydrogen-bond donor table (mdx_set_hbond_detection)
    h->h_bond_pairs.assign(s->bond_idx ? s->bond_idx : nullptr, s->bond_idx ? s->bond_idx + 2 * (size_t)s->n_bonds : nullptr);
    if (s->constraint_idx) h->h_bond_pairs.insert(h->h_bond_pairs.end(), s->constraint_idx, s->constraint_idx + 2 * (size_t)s->n_constraints);

    // ---- dynamic state staging ----
    {
        std::vector<float4> p4(N), v4(N);
        for (uint32_t i = 0; i < N; ++i) {
            p4[i] = make_float4(s->pos[3 * i], s->pos[3 * i + 1], s->pos[3 * i + 2], 0.f);
            v4[i] = s->vel ? make_float4(s->vel[3 * i], s->vel[3 * i + 1], s->vel[3 * i + 2], 0.f)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            if (invm[i] == 0.f) v4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        MDX_TRY(upload_vec(&d.pos_orig, p4, st)); MDX_TRY(upload_vec(&d.vel_orig, v4, st));
        MDX_TRY(alloc_n(&d.ext_orig, N));
        std::vector<uint32_t> gid(N);
        std::vector<uint8_t> lf(N);
        for (uint32_t i = 0; i < N; ++i) { gid[i] = i; lf[i] = (h->flags[i] & MDX_ATOM_GHOST) ? 1 : 0; }
        MDX_TRY(upload_vec(&d.gid, gid, st)); MDX_TRY(upload_vec(&d.lflag, lf, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    MDX_TRY(alloc_n(&d.slot_of, N)); MDX_TRY(alloc_n(&d.cell_of, N)); MDX_TRY(alloc_n(&d.sorted_orig, N)); MDX_TRY(alloc_n(&d.sorted_tmp, N));
    MDX_TRY(alloc_n(&d.ctl, 1)); MDX_TRY(alloc_n(&d.energy, EN_COUNT + 8 + MDX_ESTRIDE * MDX_EPART)); MDX_TRY(alloc_n(&d.flags_dev, 4));
    HIP_TRY(hipMemsetAsync(d.ctl, 0, sizeof(StepCtl), st));
    HIP_TRY(hipMemsetAsync(d.energy, 0, sizeof(double) * (EN_COUNT + 8 + MDX_ESTRIDE * MDX_EPART), st));
    HIP_TRY(hipHostMalloc((void**)&h->h_ctl, sizeof(StepCtl) + 64, hipHostMallocDefault));   // + the sequence word of ctl_readback_kernel
    std::memset(h->h_ctl, 0, sizeof(StepCtl) + 64);
    HIP_TRY(hipStreamSynchronize(st));  // host vectors go out of scope
    h->in_slot_space = false; h->list_valid = false; h->forces_valid = false;
    MDX_TRY(mdx_build_constraints(h, s));
    MDX_TRY(mdx_build_ewald_table(h));
    h->cons_dirty = mdx_has_constraints(h);
    {   // path split (DeviceState): handles whose atoms only the drift passes (and, decomposed, the halo unpack) move - the
        // constraint solvers and the one-pass water step keep the accumulator in ref[].w.  MDX_PATH_SPLIT=0: A/B
        const char* e = std::getenv("MDX_PATH_SPLIT");
        h->path_split = !mdx_has_constraints(h) && h->n_vsites == 0 && !(e && e[0] == '0');
    }
    MDX_TRY(mdx_pme_setup(h));
    MDX_TRY(mdx_rebuild(h));
    return MDX_OK;
}

extern "C" int mdx_create(const mdx_system* sys, const mdx_config* cfg, int device, mdx_handle** out) {
    if (!out) FAIL(MDX_EPARAM, "null out pointer");
    *out = nullptr;
    MDX_TRY(validate(sys, cfg));
    mdx_handle* h = new (std::nothrow) mdx_handle();
    if (!h) FAIL(MDX_EOOM, "host allocation failed");
    int rc = create_impl(sys, cfg, device, h);
    if (rc != MDX_OK) { std::string keep = g_last_error; mdx_destroy(h); g_last_error = keep; return rc; }
    *out = h;
    return MDX_OK;
}

// ---------------------------------------------------------------------------------------------
static uint32_t stale_threshold_bits(const mdx_handle* h) {
    if (std::isinf(h->r_list)) return f2u(1.0e29f);  // all pairs listed: never stale (NaN still trips)
    const float half = 0.5f * h->cfg.skin;
    return f2u(half * half);
}

// End of a chunk: the step-control words reach the host.  A kernel copies them into pinned host memory and writes a
// sequence word last; the host spins on that word - no runtime call on the wait path (tools/ubench/sync_latency.hip:
// launch + readback + wait 13 us this way, 18 us with hipStreamSynchronize behind the same kernel, 27 us with a
// hipMemcpyAsync + hipStreamSynchronize).  One chunk end per 16 steps: 1 % of a 55 us step at 23 k atoms.  Decomposed
// handles keep the synchronising copy (their host code after the chunk relies on an idle stream), and so does a wait that
// lasts longer than 20 ms (a fault then surfaces through hipStreamSynchronize).  (Round 4: profiled handles spin too - the
// event pairs are read after the wait, when all of them have completed - and so do decomposed ones, with the drift probe of
// mdx_dd_chunk_end_probe in the same read-back; MDX_DD_CHUNK_SPIN=0 restores their synchronising copy.)
__global__ __launch_bounds__(256) void ctl_readback_kernel(const uint32_t* __restrict__ src, volatile uint32_t* dst, uint32_t nwords,
                                                           volatile uint32_t* seq_word, uint32_t seq, const uint32_t* __restrict__ extra) {
    for (uint32_t i = threadIdx.x; i < nwords; i += 256) dst[i] = src[i];
    if (threadIdx.x == 0 && extra) seq_word[1] = *extra;      // (decomposed handle: the drift probe's word, next to the sequence word)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *seq_word = seq;
}
static int ctl_to_host(mdx_handle* h) {
    hipStream_t st = h->stream;
    static const bool spin_ok = [] { const char* e = std::getenv("MDX_CHUNK_SPIN"); return !(e && e[0] == '0'); }();
    static const bool dd_spin = [] { const char* e = std::getenv("MDX_DD_CHUNK_SPIN"); return !(e && e[0] == '0'); }();
    const uint32_t* probe = nullptr;
    if (h->dd) MDX_TRY(mdx_dd_chunk_end_probe(h, &probe));
    if (!spin_ok || (h->dd && !dd_spin)) {
        HIP_TRY(hipMemcpyAsync(h->h_ctl, h->d.ctl, sizeof(StepCtl), hipMemcpyDeviceToHost, st));
        if (probe) HIP_TRY(hipMemcpyAsync(&h->dd->spec_bits, probe, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (probe) h->dd->spec_valid = true;
        return MDX_OK;
    }
    volatile uint32_t* seq_word = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<char*>(h->h_ctl) + sizeof(StepCtl));
    const uint32_t seq = ++h->ctl_seq;
    hipLaunchKernelGGL(ctl_readback_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(h->d.ctl),
                       reinterpret_cast<volatile uint32_t*>(h->h_ctl), (uint32_t)(sizeof(StepCtl) / 4), seq_word, seq, probe);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *seq_word != seq; ++spins) {
        if ((spins & 0xFFFu) == 0xFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(h->dd ? 2000 : 20)) {
            HIP_TRY(hipStreamSynchronize(st));
            if (*seq_word != seq) { mdx_set_error("internal: step-control readback did not arrive"); return MDX_EDEVICE; }
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (probe) { h->dd->spec_bits = seq_word[1]; h->dd->spec_valid = true; }
    // (a decomposed handle's host code behind the chunk - transports, collectives - synchronises the stream itself where it needs to)
    return MDX_OK;
}

static int compute_forces(mdx_handle* h, bool energy, const uint32_t* gate, uint32_t thr) {
    MdxRange range_forces(energy ? "mdx forces+energies" : "mdx forces");
    if (energy) h->stats.energy_evaluations++;
    MDX_TRY(mdx_launch_vsite_construct(h, gate, thr));     // massless sites follow their parents
    bool split = false;
    if (h->dd && h->dd->halo_pending) {                    // decomposed handle, step loop: ghost positions travel now
        h->dd->halo_pending = false;
        // tiles whose lists involve no ghost run on the side stream while the message is packed, sent and unpacked on
        // this one; the tiles that need ghosts follow the unpack
        split = !energy && mdx_dd_split_now(h);
        if (split) {
            if (mdx_nb_half(h) && !h->force_zeroed) HIP_TRY(hipMemsetAsync(h->d.force, 0, sizeof(float4) * (size_t)h->S, h->stream));
            h->force_zeroed = true;
            HIP_TRY(hipEventRecord(h->dd->ev_fork, h->stream));
            HIP_TRY(hipStreamWaitEvent(h->dd->side_stream, h->dd->ev_fork, 0));
            hipStream_t main_stream = h->stream;
            h->stream = h->dd->side_stream;
            const int rc_int = mdx_launch_nonbonded(h, energy, gate, thr, 1);
            h->stream = main_stream;
            MDX_TRY(rc_int);
            HIP_TRY(hipEventRecord(h->dd->ev_interior, h->dd->side_stream));
        }
        {
            MdxRange range_halo("mdx halo exchange");
            MDX_TRY(mdx_dd_halo_begin(h));                 // pack + ncclSend/ncclRecv group
            MDX_TRY(mdx_dd_halo_end(h));                   // unpack: ghost positions, the peers' flag words
        }
    }
    if (h->pme_on && h->pme_overlap) {                     // SPME reciprocal space on its side stream, beside the pair kernel
        MDX_TRY(mdx_pme_fork(h));
        MDX_TRY(mdx_launch_pme(h, energy, gate, thr));
    }
    MDX_TRY(mdx_launch_nonbonded(h, energy, gate, thr, split ? 2 : 0));
    // half-shell decomposition: the forces this rank computed on its ghosts go back to their owners (the bonded gather, which
    // only touches owned rows, runs while the message is on the wire when the communication stream is a separate one)
    const int fr_word = (h->dd && gate) ? h->nb_step + 1 : -1;
    if (h->dd) { MdxRange range_fr("mdx ghost-force return"); MDX_TRY(mdx_dd_force_return_begin(h, fr_word)); }
    // the interior tiles join here: they touch owned rows only (tile_class_kernel), the message above carries ghost rows only,
    // so the ~10 us event hop of the join passes behind the pack and the send instead of in front of them
    if (split) HIP_TRY(hipStreamWaitEvent(h->stream, h->dd->ev_interior, 0));
    MDX_TRY(mdx_launch_bonded(h, energy, gate, thr));
    if (h->dd) MDX_TRY(mdx_dd_force_return_end(h, fr_word));
    if (h->pme_on && h->pme_overlap) MDX_TRY(mdx_pme_join(h, gate, thr));
    else MDX_TRY(mdx_launch_pme(h, energy, gate, thr));    // SPME reciprocal space (hipFFT), if requested
    MDX_TRY(mdx_launch_vsite_spread(h, gate, thr));        // ... and hand their force back to them
    MDX_TRY(mdx_launch_add_ext(h, gate, thr));
    return MDX_OK;
}

int mdx_compute_forces(mdx_handle* h, bool energy, const uint32_t* gate, uint32_t thr) {
    return compute_forces(h, energy, gate, thr);
}

static int ensure_ready(mdx_handle* h);
int mdx_ensure_ready(mdx_handle* h) { return ensure_ready(h); }

static int ensure_ready(mdx_handle* h) {
    HIP_TRY(hipSetDevice(h->device));
    if (!h->list_valid) MDX_TRY(mdx_rebuild(h));
    if (h->cons_dirty) {   // coordinates came from outside: project them onto the constraints once
        MDX_TRY(mdx_launch_constrain_positions(h, 0.f, nullptr, nullptr, 0));
        MDX_TRY(mdx_launch_constrain_velocities(h, nullptr, 0));
        h->cons_dirty = false; h->forces_valid = false; h->moved_outside = true;
        if (h->dd && h->dd->world > 1) {     // the owners projected their clusters: the ghost copies follow by message
            h->dd->halo_step = -1;
            MDX_TRY(mdx_dd_halo_begin(h));
            MDX_TRY(mdx_dd_halo_end(h));
        }
    }
    if (!h->forces_valid) {
        MDX_TRY(compute_forces(h, false, nullptr, 0));
        h->forces_valid = true;
    }
    return MDX_OK;
}

// ---- rocprof markers -------------------------------------------------------------------------
namespace {
struct RoctxApi { int state = 0; int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };   // state: 0 unknown, 1 on, 2 off
RoctxApi g_roctx;
std::mutex g_roctx_mutex;
bool roctx_ready() {
    if (g_roctx.state) return g_roctx.state == 1;
    std::lock_guard<std::mutex> lk(g_roctx_mutex);
    if (g_roctx.state) return g_roctx.state == 1;
    const char* e = std::getenv("MDX_ROCTX");
    int st = 2;
    if (e && e[0] == '1') {
        const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
        for (const char* n : names) {
            void* lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) continue;
            *(void**)(&g_roctx.push) = dlsym(lib, "roctxRangePushA");
            *(void**)(&g_roctx.pop) = dlsym(lib, "roctxRangePop");
            if (g_roctx.push && g_roctx.pop) { st = 1; break; }
        }
    }
    g_roctx.state = st;
    return st == 1;
}
}  // namespace
void mdx_range_push(const char* name) { if (roctx_ready()) (void)g_roctx.push(name); }
void mdx_range_pop() { if (roctx_ready()) (void)g_roctx.pop(); }

// ---- profiling -------------------------------------------------------------------------------
void mdx_prof_begin(mdx_handle* h, int kind, hipStream_t st) {
    h->prof_open = false;
    if (!h->profile) return;
    if (h->profile_level == 2 && kind != 0 && kind != 4) return;   // pair kernel of the step loop only
    if (h->profile_level == 2) {
        // ... and of those every eighth launch (MDX_PROF_SAMPLE; 1 = all): two hipEventRecords around a launch cost 10-14 us of a
        // step - 28 % of a 23 k-atom step, 16 % at 51 k atoms, 2 % at 1 M (bench.py --profile-level 0 against 2, round 4) - so a
        // bracket on every launch made the measured rate an artefact of measuring.  The two halves of a split launch go together.
        static const uint32_t every = [] { const char* e = std::getenv("MDX_PROF_SAMPLE"); const int v = e ? std::atoi(e) : 8; return (uint32_t)(v < 1 ? 1 : v); }();
        if (kind == 0) h->prof_sampled = (h->prof_seq++ % every) == 0u;
        if (!h->prof_sampled) return;
    }
    if (kind >= 6 && h->profile_level != 3) return;                // the phases of a decomposed step: level 3
    h->prof_open = true;
    mdx_handle::EvPair p{};
    p.kind = kind; p.tag = h->prof_tag; p.st = st ? st : h->stream;
    hipEvent_t* ev[2] = {&p.a, &p.b};
    for (auto e : ev) {
        if (!h->ev_pool.empty()) { *e = h->ev_pool.back(); h->ev_pool.pop_back(); }
        else (void)hipEventCreate(e);
    }
    (void)hipEventRecord(p.a, p.st);
    h->ev_pending.push_back(p);
}
void mdx_prof_end(mdx_handle* h) {
    if (!h->profile || !h->prof_open || h->ev_pending.empty()) return;
    h->prof_open = false;
    (void)hipEventRecord(h->ev_pending.back().b, h->ev_pending.back().st);
}
void mdx_prof_collect(mdx_handle* h, int first_stale_step) {
    for (auto& p : h->ev_pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(p.b);
        // launches enqueued behind a stale list were no-ops: the integrate pass of the stale step
        // itself still ran (it detected it), force kernels of that step and everything later did not
        const bool ran = p.tag < 0 || ((p.kind == 2 || p.kind == 5) ? p.tag <= first_stale_step : p.tag < first_stale_step);
        // (level 2 = "the pair kernel only" covers both halves of a split launch)
        if (ran && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            if (p.kind == 0) { h->stats.nb_ms_sum += ms; h->stats.nb_launches++; }
            else if (p.kind == 4) h->stats.nb_ms_sum += ms;     // the boundary half of a split pair-kernel launch
            else if (p.kind == 1) { h->stats.bonded_ms_sum += ms; h->stats.bonded_launches++; }
            else if (p.kind == 2) { h->stats.integ_ms_sum += ms; h->stats.integ_launches++; }
            else if (p.kind == 5) { h->stats.fused_ms_sum += ms; h->stats.fused_launches++; }
            if (h->dd && h->profile_level == 3) {      // mdx_comm_diag: phase by phase
                int ph = -1;
                if (p.kind >= 6 && p.kind <= 11) ph = p.kind - 6;
                else if (p.kind == 0) ph = MDX_PHASE_PAIR;
                else if (p.kind == 4) ph = MDX_PHASE_PAIR_BOUNDARY;
                else if (p.kind == 1 || p.kind == 5) ph = MDX_PHASE_BONDED;
                else if (p.kind == 2) ph = MDX_PHASE_INTEGRATE;
                if (ph >= 0) { h->dd->phase_ms[ph] += ms; h->dd->phase_n[ph]++; }
            }
        }
        h->ev_pool.push_back(p.a); h->ev_pool.push_back(p.b);
    }
    h->ev_pending.clear();
}

// mdx_config.skin == 0: the Verlet skin is the library's to choose.  A larger skin means fewer list rebuilds and more listed
// pairs on every step; where the balance lies depends on the time step, the temperature and on what else the step holds (at the
// reference's default operating point - rigid water, dt 2 fs - the list goes stale every 4-5 steps at 2 A).  No model: the step
// rate is MEASURED over windows of >= 12 rebuilds (>= 400 steps) at the current skin, then at 0.5 A less; while that is > 3 %
// faster the walk goes on downwards in 0.25 A steps (not below 0.75 A), else it tries upwards in 0.5 A steps while the box
// allows (edge >= 2 (cutoff + skin)) and the rate does not fall; the best stays if it beats the start by 1 %.  Forces do not depend on the skin, so nothing but speed changes.  Single
// device only: ranks of a decomposed handle would have to agree, and their clocks do not.
static void skin_apply(mdx_handle* h, float s) {
    h->cfg.skin = s;
    h->r_list = std::max(h->cfg.lj_cutoff, h->cfg.coulomb_cutoff) + s;
    h->list_valid = false;
    h->inner_skin_auto = 0.f; h->dual_auto_off = false; h->dual_win_steps = 0; h->dual_win_prunes = 0;
    h->stretch_samples = 0;
    h->skin_tune.win_steps = 0; h->skin_tune.win_rebuilds = 0; h->skin_tune.skip_rebuilds = 5;     // (the first lists at a new radius may regrow their arrays; the chunk-length prediction relearns over three intervals)
}
static bool skin_fits(const mdx_handle* h, float s) {
    for (int d = 0; d < 3; ++d)
        if (h->per[d] && h->box_hi[d] - h->box_lo[d] < 2.0f * (std::max(h->cfg.lj_cutoff, h->cfg.coulomb_cutoff) + s)) return false;
    return true;
}
static void skin_autotune(mdx_handle* h, uint32_t done, bool stale_hit) {
    mdx_handle::SkinTune& t = h->skin_tune;
    const auto now = std::chrono::steady_clock::now();
    if (t.phase == 0) {      // warm-up: lists, arrays and the dual-list tuning settle
        t.warm_steps += done;
        if (t.warm_steps >= 200) { t.phase = 1; t.win_steps = 0; t.win_rebuilds = 0; t.skip_rebuilds = 1; }
        return;
    }
    if (t.skip_rebuilds) { if (stale_hit && --t.skip_rebuilds == 0) { t.t0 = now; t.win_steps = 0; t.win_rebuilds = 0; } return; }
    t.win_steps += done; t.win_rebuilds += stale_hit ? 1u : 0u;
    if (!((t.win_rebuilds >= 12 && t.win_steps >= 400 && stale_hit) || t.win_steps >= 6000)) return;     // (windows end ON a rebuild: whole stretches)
    const double rate = (double)t.win_steps / std::chrono::duration<double>(now - t.t0).count();
    const float cur = h->cfg.skin;
    static const bool dbg = [] { const char* e = std::getenv("MDX_DEBUG_SKIN"); return e && e[0] == '1'; }();
    if (dbg) fprintf(stderr, "[mdx skin tune] phase %d skin %.2f: %.1f steps/s over %u steps, %u rebuilds (best %.1f at %.2f, base %.1f)\n", t.phase, cur, rate, t.win_steps, t.win_rebuilds, t.best_rate, t.best_skin, t.base_rate);
    auto finish = [&]() { t.phase = 4; if (h->cfg.skin != t.best_skin) skin_apply(h, t.best_skin); };
    if (t.phase == 1) {
        t.base_rate = t.best_rate = rate; t.base_skin = t.best_skin = cur;
        if (cur - 0.5f >= 0.75f) { t.phase = 2; skin_apply(h, cur - 0.5f); }
        else if (skin_fits(h, cur + 0.5f)) { t.phase = 3; skin_apply(h, cur + 0.5f); }
        else t.phase = 4;
    } else if (t.phase == 2) {
        if (rate > t.best_rate * 1.03) {
            t.best_rate = rate; t.best_skin = cur;
            if (cur - 0.25f >= 0.75f) skin_apply(h, cur - 0.25f); else finish();
        } else if (t.best_skin == t.base_skin && skin_fits(h, t.base_skin + 0.5f)) { t.phase = 3; skin_apply(h, t.base_skin + 0.5f); }
        else finish();
    } else if (t.phase == 3) {
        // Upwards the rate climbs in small steps (1,048,576 OPC sites, dt 2 fs, SPME: 849 steps/s at 2.0 A, 865 at 3.0, 875 at 3.5 -
        // +1 ... +1.5 % per half Angstrom): a walk that wants 3 % per step stops at once.  It goes on while a step is not worse than the
        // best by 1 % (windows repeat to ~0.5 %), remembers the best, and keeps it if it beats the base by 1 %.
        if (rate > t.best_rate * 1.005) { t.best_rate = rate; t.best_skin = cur; }
        if (rate >= t.best_rate * 0.99 && skin_fits(h, cur + 0.5f) && cur + 0.5f <= 4.0f) skin_apply(h, cur + 0.5f);
        else {
            if (t.best_rate < t.base_rate * 1.01) { t.best_skin = t.base_skin; t.best_rate = t.base_rate; }
            finish();
        }
    }
}

// ---------------------------------------------------------------------------------------------
extern "C" int mdx_step(mdx_handle* h, float dt, const float* ext_forces, uint32_t n_steps) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    if (!std::isfinite(dt)) FAIL(MDX_EPARAM, "non-finite dt");
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    DeviceState& d = h->d;
    // external forces (held constant over the burst)  [ref: src/mol_alignment.rs:318-346]
    const bool had_ext = h->have_ext;
    if (ext_forces) {
        if (h->n_local != h->N && !h->dd) FAIL(MDX_EPARAM, "external forces need the library's own decomposition (mdx_comm_init) on a narrowed handle");
        std::vector<float4> e4(h->N);
        for (uint32_t i = 0; i < h->N; ++i)
            e4[i] = make_float4(ext_forces[3 * i], ext_forces[3 * i + 1], ext_forces[3 * i + 2], 0.f);
        HIP_TRY(hipMemcpyAsync(d.ext_orig, e4.data(), sizeof(float4) * h->N, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        h->have_ext = true; h->forces_valid = false;
    } else if (had_ext) {
        h->have_ext = false; h->forces_valid = false;
    }
    if (n_steps == 0) return MDX_OK;
    MDX_TRY(ensure_ready(h));
    // dual list: what moved the atoms since the last call without going through a list rebuild (the minimiser, a
    // constraint projection, a pose update) did not feed the path accumulators: then the first force call re-prunes the
    // inner list.  Otherwise the inner list of the last call is still exact and a 10-step GUI burst does not pay a pruning
    // pass (+0.17 ms at 1 M atoms) at its start.
    if (h->moved_outside) { h->prune_pending = true; h->moved_outside = false; }
    uint32_t thr = stale_threshold_bits(h);      // (the skin tuning below may change it between chunks)
    uint32_t remaining = n_steps;
    while (remaining) {
        MDX_TRY(ensure_ready(h));   // a barostat application at the last cadence point left the list to rebuild
        uint32_t chunk = std::min(std::min(remaining, h->cfg.chunk_steps), mdx_steps_to_next_event(h));
        // The host enqueues a chunk blind; behind the step at which the list goes stale every launch of the chunk is gated off
        // (~4 us each, three per step at 1 M atoms: ~8 steps' worth per rebuild with fixed chunks of 16).  Rebuild-free
        // stretches are regular (thermal motion against a fixed skin), so a chunk is cut to end just behind the step the
        // stretch is expected to end at: mean + mean absolute deviation + 1.  MDX_CHUNK_PREDICT=0: fixed chunks (A/B).
        static const bool chunk_predict = [] { const char* e = std::getenv("MDX_CHUNK_PREDICT"); return !(e && e[0] == '0'); }();
        if (chunk_predict && h->stretch_samples >= 3 && chunk > 4) {
            const float left = h->stretch_mean + h->stretch_dev + 1.0f - (float)h->steps_since_rebuild;
            const uint32_t want = left > 4.0f ? (uint32_t)std::ceil(left) : 4u;
            chunk = std::min(chunk, want);
        }
        const auto t_chunk = std::chrono::steady_clock::now();
        MdxRange range_chunk("mdx step chunk");
        HIP_TRY(hipMemsetAsync(d.ctl, 0, sizeof(StepCtl), st));
        // Velocity Verlet: the closing half kick of step s and the opening one of step s+1 are one pass (mode 1), with
        // and without constraints.  RATTLE's velocity projection between the two half kicks removes components along
        // the bonds at t + dt; left in, they displace the next drift along exactly the directions the next SHAKE corrects
        // along (the bonds at t + dt are its "old" bonds, the mass weighting is the same), so its multipliers absorb them:
        // the positions are the same, and SHAKE's dx / dt puts the half-step velocities back on the constraint surface.
        // The projection therefore runs once, after the closing kick of the chunk, where velocities are read (kinetic
        // energy, thermostats, download) - two launches per step less at the reference's default operating point.
        // MDX_RATTLE_EVERY_STEP=1 keeps kick-drift-SHAKE-forces-kick-RATTLE per step for A/B.
        // Leapfrog and Langevin-middle carry half-step velocities: one pass per step (mode 1 / 3), SHAKE
        // folds its correction into them, and there is no closing kick.
        static const bool rattle_every = [] { const char* e = std::getenv("MDX_RATTLE_EVERY_STEP"); return e && e[0] == '1'; }();
        const int integ = h->integrator;
        const bool vv = integ == MDX_INTEGRATOR_VERLET_VELOCITY;
        const bool fused = vv && (!mdx_has_constraints(h) || !rattle_every);
        // Large classes: steps 1 .. chunk-1 run bonded gather + full kick + drift as ONE pass over double-buffered positions
        // (mdx_integrate.hip); the force call in front of such a pass leaves its bonded launch out.  The last force call of
        // the chunk is complete again (the closing half kick, energies, downloads read the force array).
        const bool fuse_bi = fused && mdx_bonded_integrate_ok(h);
        // energies are read after this chunk (cadence, snapshot, barostat): its last force call evaluates them
        const bool want_e = mdx_energy_wanted_at(h, h->step_count + chunk);
        h->e_cache_valid = false; h->e_pending = false;
        const size_t e_bytes = sizeof(double) * (EN_COUNT + 8 + MDX_ESTRIDE * MDX_EPART);
        float4* pos_after[MDX_MAX_CHUNK + 1];      // the buffer that holds the positions after step s's drift
        // boxes of rigid water (the reference's default operating point): site-force spread + kick + drift + SETTLE + site placement in ONE pass
        const bool water = fused && mdx_water_step_ok(h);
        const bool pipe_chunk = fuse_bi && mdx_dd_fold_eligible(h);      // decomposed handle: the fused pass also packs the halo and adds the returned ghost forces
        // One launch per step (round 6; the one-wave-per-tile class): the pair launch of step s also finishes the drift of step s for its
        // tile's atoms and evaluates their bonded roles (mdx_nonbonded_impl.h STEP).  Positions travel in the step form Y between two
        // buffers, forces rotate through three (read / accumulate / zero); step s reads ybuf[s & 1] and fb3[s % 3].  A chunk that ends with
        // an energy evaluation leaves the form one step early (the energy flavour is a kernel of its own).
        const bool onepass = fused && !water && !pipe_chunk && chunk >= (want_e ? 2u : 1u) && mdx_onepass_ok(h);
        float4* const ybuf[2] = {d.posq_alt, d.posq};
        float4* const fb3[3] = {d.force, d.force_b, d.force_c};
        auto onepass_point_at = [&](uint32_t n) {      // the plain arrays after n position stages of this chunk
            d.posq = ybuf[n & 1]; d.posq_alt = ybuf[(n + 1) & 1];
            d.force = fb3[n % 3]; d.force_b = fb3[(n + 1) % 3]; d.force_c = fb3[(n + 2) % 3];
        };
        for (uint32_t s = 0; s < chunk; ++s) {
            h->prof_tag = (int)s;
            h->lang_step = h->step_count + s;
            if (onepass) {
                if (s == 0) MDX_TRY(mdx_launch_step_begin(h, dt, &d.ctl->disp2[0], &d.ctl->disp2[1], thr, &d.ctl->prune[1]));
                pos_after[s] = ybuf[(s + 1) & 1];
                if (!(want_e && s + 1 == chunk)) {
                    OnePassNow& o = h->onepass;
                    o.yin = ybuf[s & 1]; o.yout = ybuf[(s + 1) & 1];
                    o.fprev = fb3[s % 3]; o.fcur = fb3[(s + 1) % 3]; o.fnext = fb3[(s + 2) % 3];
                    o.dt = dt; o.last = s + 1 == chunk; o.first = s == 0;
                    o.disp_out = &d.ctl->disp2[s + 2]; o.prune_out = o.last ? nullptr : &d.ctl->prune[s + 2]; o.viol = &d.ctl->viol[s];
                    h->nb_step = (int)s;
                    const int frc = mdx_launch_nonbonded(h, false, &d.ctl->disp2[s + 1], thr, 0);
                    h->nb_step = -1; h->onepass = OnePassNow{};
                    MDX_TRY(frc);
                    if (h->onepass_refused) FAIL(MDX_EDEVICE, "internal: no one-launch-per-step instantiation for this handle's pair kernel");
                    if (s + 1 == chunk) onepass_point_at(chunk);
                    continue;
                }
                // the chunk's last step, energies wanted: its drift as a pass of its own (back into the plain form), then the energy flavour
                onepass_point_at(chunk);
                MDX_TRY(mdx_launch_step_materialise(h, dt, ybuf[s & 1], fb3[s % 3], ybuf[(s + 1) & 1], false, 1.0f, &d.ctl->disp2[s + 1], thr));
                HIP_TRY(hipMemsetAsync(d.energy, 0, e_bytes, st));
                h->nb_step = (int)s;
                // (the buffer this evaluation accumulates into was zeroed by the launch before it; the fill the pair launch would enqueue
                // is not gated - behind a stale step it would wipe force rows the way back out of the step form still reads)
                h->force_zeroed = true;
                const int frc = compute_forces(h, true, &d.ctl->disp2[s + 1], thr);
                h->nb_step = -1;
                MDX_TRY(frc);
                continue;
            }
            const int mode = !vv ? (integ == MDX_INTEGRATOR_LANGEVIN_MIDDLE ? 3 : 1) : ((s == 0 || !fused) ? 0 : 1);
            if (h->dd) h->dd->pipe_now = pipe_chunk && mode == 1 && !(want_e && s + 1 == chunk);
            const bool water_now = water && (mode == 0 || mode == 1);
            const bool water_rest = water_now && mdx_water_step_mixed(h);      // a solute beside the waters: it keeps its own passes
            if (water_now) {
                MDX_TRY(mdx_launch_water_step(h, mode, dt, &d.ctl->disp2[s], &d.ctl->disp2[s + 1], thr, &d.ctl->prune[s + 1]));
                if (water_rest) MDX_TRY(mdx_launch_integrate(h, mode, dt, &d.ctl->disp2[s], &d.ctl->disp2[s + 1], thr, &d.ctl->prune[s + 1], true));
            }
            else if (fuse_bi && mode == 1) MDX_TRY(mdx_launch_bonded_integrate(h, dt, &d.ctl->disp2[s], &d.ctl->disp2[s + 1], thr, &d.ctl->prune[s + 1]));
            else MDX_TRY(mdx_launch_integrate(h, mode, dt, &d.ctl->disp2[s], &d.ctl->disp2[s + 1], thr, &d.ctl->prune[s + 1]));
            pos_after[s] = d.posq;
            // Rigid-water boxes under Ewald: the only "bonded" terms are the exclusion corrections between the sites of one rigid
            // molecule - a self-equilibrated, in-plane force system on a rigid triangle, i.e. a combination of pair forces along its
            // three constrained edges at time t, which is exactly the form of a SHAKE / SETTLE correction: the constrained positions
            // and velocities come out the same with or without them (the multipliers absorb them).  Force calls whose result only
            // the next step's kick reads leave them out; the last two of a chunk are complete (forces, energies and the constraint
            // virial of the chunk's last position stage are then what they always were).  MDX_RIGID_EXCL_SKIP=0: A/B.
            static const bool excl_skip_env = [] { const char* e = std::getenv("MDX_RIGID_EXCL_SKIP"); return !(e && e[0] == '0'); }();
            const bool excl_skip = excl_skip_env && water && h->wstep_all && h->excl_inside_rigid && h->n_roles != 0 && h->n_roles == h->n_roles_excl && s + 2 < chunk &&
                                   !(want_e && s + 1 == chunk);
            h->bonded_deferred = (fuse_bi && s + 1 < chunk) || excl_skip;
            if (!water_now || water_rest) {
                h->cons_full_kick = vv && mode == 1;
                MDX_TRY(mdx_launch_constrain_positions(h, dt, &d.ctl->disp2[s], &d.ctl->disp2[s + 1], thr, &d.ctl->prune[s + 1], water_now));
                h->cons_full_kick = false;
            }
            // Langevin middle: friction and noise sit between the two half drifts, so SHAKE's dx/dt is not an
            // exact velocity projection; RATTLE the half-step velocities (same gate as SHAKE: it belongs to the
            // drift, which has happened even when the step's forces turn out to be gated off)
            if (integ == MDX_INTEGRATOR_LANGEVIN_MIDDLE) MDX_TRY(mdx_launch_constrain_velocities(h, &d.ctl->disp2[s], thr));
            h->nb_step = (int)s;      // the pair kernel of this call may walk the inner masks / prune (prune[s + 1])
            if (h->dd) { h->dd->halo_pending = true; h->dd->halo_step = (int)s; h->chunk_s = (int)s; }
            const bool e_now = want_e && s + 1 == chunk;
            if (e_now) HIP_TRY(hipMemsetAsync(d.energy, 0, e_bytes, st));
            h->vsite_spread_deferred = water && h->wstep_sites_all && s + 1 < chunk && !e_now;      // (the next step's water_step_kernel spreads the sites' forces)
            const int frc = compute_forces(h, e_now, &d.ctl->disp2[s + 1], thr);
            h->nb_step = -1; h->bonded_deferred = false; h->vsite_spread_deferred = false;
            MDX_TRY(frc);
            if (vv && !fused) {
                MDX_TRY(mdx_launch_integrate(h, 2, dt, &d.ctl->disp2[s + 1], nullptr, thr));
                MDX_TRY(mdx_launch_constrain_velocities(h, &d.ctl->disp2[s + 1], thr));
            }
        }
        h->prof_tag = (int)chunk;
        if (fused) {
            // (one launch per step, no energy step at the end: the last launch passes its own gate - or a kick beyond its grant - on in the next word)
            MDX_TRY(mdx_launch_integrate(h, 2, dt, &d.ctl->disp2[(onepass && !want_e) ? chunk + 1 : chunk], nullptr, thr));
            MDX_TRY(mdx_launch_constrain_velocities(h, &d.ctl->disp2[chunk], thr));     // (no-op without constraints)
        }
        h->prof_tag = -1;
        if (h->dd) { h->chunk_s = -1; mdx_dd_pipe_chunk_end(h); }
        // MDX_DEBUG_HOST=1: where the host spends a chunk - enqueueing it, or waiting for the device to finish it
        static const bool dbg_host = [] { const char* e = std::getenv("MDX_DEBUG_HOST"); return e && e[0] == '1'; }();
        const auto t_enq = std::chrono::steady_clock::now();
        if (h->dd) h->dd->probe_both_buffers = fuse_bi && chunk > 1;
        MDX_TRY(ctl_to_host(h));
        if (h->dd) h->dd->probe_both_buffers = false;
        if (dbg_host && chunk >= 8) {      // (chunks cut short by a per-step thermostat or cadence say nothing about the step loop)
            static double enq = 0.0, wait = 0.0; static unsigned long long steps = 0, chunks = 0;
            const auto t_w = std::chrono::steady_clock::now();
            enq += std::chrono::duration<double, std::micro>(t_enq - t_chunk).count();
            wait += std::chrono::duration<double, std::micro>(t_w - t_enq).count();
            steps += chunk; ++chunks;
            if ((chunks & 31ull) == 0ull)
                std::fprintf(stderr, "[mdx host] %llu steps in %llu chunks: enqueue %.1f us per step, wait at the chunk end %.1f us per step\n",
                             steps, chunks, enq / (double)steps, wait / (double)steps);
        }
        uint32_t done = chunk;
        bool stale_hit = false;
        int first_stale = 1 << 30;
        for (uint32_t s = 0; s < chunk; ++s)
            if (h->h_ctl->disp2[s + 1] > thr || (onepass && h->h_ctl->viol[s])) { first_stale = (int)s; break; }
        if (h->profile) mdx_prof_collect(h, first_stale);
        static const bool dbg_stale = [] { const char* e = std::getenv("MDX_DEBUG_STALE"); return e && e[0] == '1'; }();
        if (dbg_stale && first_stale < (1 << 30))
            std::fprintf(stderr, "[mdx] list stale at step %d of a chunk of %u (step count %llu): word %.4g against (skin / 2)^2 = %.4g%s\n", first_stale, chunk,
                         (unsigned long long)h->step_count, (double)u2f(h->h_ctl->disp2[first_stale + 1]), (double)u2f(thr), h->dd ? " [decomposed]" : "");
        for (uint32_t s = 0; s < chunk; ++s) {
            const bool kick_beyond_grant = onepass && h->h_ctl->viol[s] != 0u;
            if (h->h_ctl->disp2[s + 1] > thr || kick_beyond_grant) {
                // (fused bonded + kick + drift passes swap the two position buffers at every enqueued step, the gated-off
                // ones included: the state is in the buffer step s's drift wrote - also for a caller that downloads the
                // coordinates to look at a runaway step)
                if (fuse_bi && !onepass && d.posq != pos_after[s]) std::swap(d.posq, d.posq_alt);
                if (onepass) {
                    // the launch of step s stayed a no-op (or its kick went beyond what the words had granted: its forces are not to be
                    // trusted, its kick is done): the drift of step s out of the step form, into the buffer that launch would have written
                    onepass_point_at(s + 1);
                    MDX_TRY(mdx_launch_step_materialise(h, dt, ybuf[s & 1], fb3[s % 3], ybuf[(s + 1) & 1], kick_beyond_grant, s == 0 ? 0.5f : 1.0f, nullptr, 0));
                    if (kick_beyond_grant) ++h->onepass_violations;
                }
                if (u2f(h->h_ctl->disp2[s + 1]) > 1.0e29f) {
                    h->forces_valid = false; h->list_valid = false;
                    h->step_count += s;
                    FAIL(MDX_ENAN, "non-finite or runaway coordinates during mdx_step");
                }
                // the drift of step s happened, its forces did not: rebuild, finish the step.  On a decomposed handle
                // every rank is here at the same step (the flag rides on the halo message): rebuild locally while the
                // owned + ghost set is still complete, else repartition - decided alike on every rank
                h->list_valid = false;
                const float len = (float)(h->steps_since_rebuild + s + 1);   // (mdx_rebuild restarts the count)
                if (h->dd) { h->chunk_s = -1; MDX_TRY(mdx_dd_on_stale(h)); }
                else MDX_TRY(mdx_rebuild(h));
                {   // length of the stretch that just ended (running mean and mean absolute deviation)
                    if (h->stretch_samples == 0) { h->stretch_mean = len; h->stretch_dev = 2.0f; }
                    else {
                        h->stretch_dev += 0.25f * (std::fabs(len - h->stretch_mean) - h->stretch_dev);
                        h->stretch_mean += 0.25f * (len - h->stretch_mean);
                    }
                    ++h->stretch_samples;
                    stale_hit = true;
                }
                const bool e_now = want_e && s + 1 == chunk;
                if (e_now) HIP_TRY(hipMemsetAsync(d.energy, 0, e_bytes, st));
                h->nb_post_rebuild = !e_now;     // dual list: this force call is the pruning pass over the new list
                const int frc = compute_forces(h, e_now, nullptr, 0);
                h->nb_post_rebuild = false;
                MDX_TRY(frc);
                if (vv) {
                    MDX_TRY(mdx_launch_integrate(h, 2, dt, nullptr, nullptr, thr));
                    MDX_TRY(mdx_launch_constrain_velocities(h, nullptr, 0));
                }
                done = s + 1;
                break;
            }
        }
        if (h->dd) h->dd->spec_valid = false;      // (the chunk-end drift probe speaks for the state at THIS chunk's end only)
        // Dual list, library-default buffer: keep the pruning passes rare enough to pay.  A pass costs +15 % over the plain
        // list and an inner-list step saves 15 %, so above ~45 % passes the dual list loses.  Long steps (dt = 2 fs with
        // constraints moves atoms four times as far per step as the 0.5 fs of flexible water) hit the 0.25 A path budget
        // every other step: the buffer grows by 0.25 A while more than 30 % of the steps prune, and when there is no room
        // left under the Verlet skin and the passes (the forced one behind every rebuild included) still exceed 30 %, the handle goes
        // back to the plain list.
        if (h->dual_on && h->cfg.inner_skin == 0.f) {
            for (uint32_t s = 0; s < done; ++s) h->dual_win_prunes += h->h_ctl->prune[s + 1] != 0u;
            // (round 3: the force call behind a rebuild is a pruning pass too; at dt = 2 fs there is a rebuild every 4-5 steps)
            if (stale_hit && !h->inner_from_rebuild) ++h->dual_win_prunes;
            h->dual_win_steps += done;
            if (h->dual_win_steps >= 96) {
                const float f = (float)h->dual_win_prunes / (float)h->dual_win_steps;
                if (f > 0.30f) {
                    if (h->inner_skin + 0.25f <= h->cfg.skin - 0.75f) {
                        h->inner_skin_auto = h->inner_skin + 0.25f; h->inner_skin = h->inner_skin_auto;
                        h->prune_pending = true;       // the next force call prunes with the new radius: paths restart there
                    } else {
                        // no room left under the Verlet skin: at the largest buffer the inner list keeps 84 % of the cluster
                        // pairs, and with three passes in ten steps the plain list measured 1-3 % faster (round 3: the
                        // reference's default operating point, rigid OPC at dt = 2 fs; the threshold was 42 % before)
                        h->dual_auto_off = true; h->dual_on = false;
                    }
                }
                h->dual_win_steps = 0; h->dual_win_prunes = 0;
            }
        }
        if (h->dd && h->dd->world > 1 && h->dd->tune_phase < 2 && h->tile_split) {
            // interior / boundary split: try both over the first full, rebuild-free chunks and keep the faster
            MdxDecomp* dd = h->dd;
            if (done == chunk && chunk == h->cfg.chunk_steps) {
                if (dd->tune_chunks >= 2) {   // (the first two chunks of a phase warm it up)
                    dd->tune_ms[dd->tune_phase] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_chunk).count();
                    dd->tune_steps[dd->tune_phase] += chunk;
                }
                if (++dd->tune_chunks >= 8) {
                    dd->tune_chunks = 0;
                    if (dd->tune_phase == 0) { dd->tune_phase = 1; dd->overlap = false; }
                    else {
                        const double with_split = dd->tune_ms[0] / std::max(dd->tune_steps[0], 1u), without = dd->tune_ms[1] / std::max(dd->tune_steps[1], 1u);
                        dd->overlap = with_split < without;
                        dd->tune_phase = 2;
                    }
                }
            }
        }
        if (h->skin_tune.on && !h->dd) {
            // the best skin depends on the time step (how many steps a list lasts): a run that relaxes at 1 fs and produces at 2 fs
            // tunes again when the step it is called with has changed by more than a fifth
            mdx_handle::SkinTune& tn = h->skin_tune;
            if (tn.dt_tuned > 0.f && fabsf(dt - tn.dt_tuned) > 0.2f * tn.dt_tuned) { tn.phase = 0; tn.warm_steps = 0; }
            tn.dt_tuned = dt;
            if (tn.phase < 4) { skin_autotune(h, done, stale_hit); thr = stale_threshold_bits(h); }
        }
        h->steps_since_rebuild = stale_hit ? 0u : h->steps_since_rebuild + done;
        h->forces_valid = true;
        h->e_pending = want_e && done == chunk;
        h->step_count += done;
        h->time_ps += (double)dt * done;
        remaining -= done;
        MDX_TRY(mdx_after_steps(h, dt, done));
    }
    h->stats.wall_ms_sum +=
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MDX_OK;
}

extern "C" uint64_t mdx_step_count(const mdx_handle* h) { return h ? h->step_count : 0; }

extern "C" int mdx_energy(mdx_handle* h, mdx_energies* out) { return mdx_energy_impl(h, out); }

static int energy_tail(mdx_handle* h, mdx_energies* out);

int mdx_finalize_energy_cache(mdx_handle* h) {
    if (!h->e_pending) return MDX_OK;
    h->e_pending = false;
    if (!h->forces_valid || !h->list_valid || h->cons_dirty) return MDX_OK;   // something moved the atoms since: evaluate afresh when asked
    // (the thermostat's own kinetic-energy launch added into the same word; the force maximum next to it is idempotent)
    HIP_TRY(hipMemsetAsync(h->d.energy + EN_KIN, 0, sizeof(double), h->stream));
    MDX_TRY(energy_tail(h, &h->e_cache));
    h->e_cache_valid = true; h->e_cache_step = h->step_count;
    return MDX_OK;
}

int mdx_energy_impl(mdx_handle* h, mdx_energies* out) {
    if (!h || !out) FAIL(MDX_EPARAM, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->e_cache_valid && h->e_cache_step == h->step_count && h->forces_valid && h->list_valid && !h->cons_dirty) {
        *out = h->e_cache;      // the step loop evaluated them with the forces of this step (mdx_set_energy_cadence)
        h->stats.energies_from_step_loop++;
        return MDX_OK;
    }
    if (!h->list_valid) MDX_TRY(mdx_rebuild(h));
    if (h->cons_dirty) MDX_TRY(ensure_ready(h));
    hipStream_t st = h->stream;
    HIP_TRY(hipMemsetAsync(h->d.energy, 0, sizeof(double) * (EN_COUNT + 8 + MDX_ESTRIDE * MDX_EPART), st));
    MDX_TRY(compute_forces(h, true, nullptr, 0));
    h->forces_valid = true;
    MDX_TRY(energy_tail(h, out));
    if (h->profile) mdx_prof_collect(h);
    return MDX_OK;
}

// kinetic energy + constraint virial behind a force call of the energy flavour, read-back, totals
static int energy_tail(mdx_handle* h, mdx_energies* out) {
    hipStream_t st = h->stream;
    MDX_TRY(mdx_launch_kinetic(h));
    MDX_TRY(mdx_launch_constraint_virial(h));   // SHAKE forces of the last step (0 after a dt = 0 projection)
    double e[EN_COUNT + 8 + MDX_ESTRIDE * MDX_EPART];
    HIP_TRY(hipMemcpyAsync(e, h->d.energy, sizeof(e), hipMemcpyDeviceToHost, st));
    if (h->wait_hook) { auto fn = h->wait_hook; h->wait_hook = nullptr; fn(h->wait_hook_arg); }
    HIP_TRY(hipStreamSynchronize(st));
    double u_cross = 0.0, du_dl = 0.0;
    for (int k = 0; k < MDX_EPART; ++k) {   // the pair kernel's partial sums
        const double* q = e + EN_COUNT + 8 + MDX_ESTRIDE * k;
        e[EN_LJ] += q[0]; e[EN_COUL] += q[1]; e[EN_VIRIAL] += q[2]; u_cross += q[3]; du_dl += q[4];
    }
    if (h->dd) {   // every rank evaluated its share (a pair's energy is split between the owners of its atoms): sum them
        double v[EN_COUNT + 2];
        for (int k = 0; k < EN_COUNT; ++k) v[k] = e[k];
        v[EN_COUNT] = u_cross; v[EN_COUNT + 1] = du_dl;
        MDX_TRY(mdx_dd_allreduce_host(h, v, EN_COUNT + 2));
        for (int k = 0; k < EN_COUNT; ++k) e[k] = v[k];
        u_cross = v[EN_COUNT]; du_dl = v[EN_COUNT + 1];
        uint32_t mfb; std::memcpy(&mfb, &e[EN_COUNT], 4);
        double mv = (double)mfb;
        MDX_TRY(mdx_dd_allreduce_host(h, &mv, 1, true));
        mfb = (uint32_t)mv; std::memcpy(&e[EN_COUNT], &mfb, 4);
    }
    std::memset(out, 0, sizeof(*out));
    out->bond = e[EN_BOND]; out->angle = e[EN_ANGLE]; out->dihedral = e[EN_DIHEDRAL];
    out->lj = e[EN_LJ]; out->coulomb = e[EN_COUL]; out->lj14 = e[EN_LJ14]; out->coulomb14 = e[EN_COUL14];
    out->kinetic = e[EN_KIN];
    out->potential_bonded = out->bond + out->angle + out->dihedral;
    out->coulomb_recip = h->pme_on ? e[EN_RECIP] + h->ewald_self + h->ewald_background : 0.0;
    out->potential_nonbonded = out->lj + out->coulomb + out->lj14 + out->coulomb14 + out->coulomb_recip;
    out->potential = out->potential_bonded + out->potential_nonbonded;
    const double dof = mdx_dof(h);
    out->temperature = 2.0 * out->kinetic / (dof * MDX_KB);
    if (h->periodic || h->dd) {   // (a rank of a fully cut box is not periodic locally: the box is still the global one)
        out->volume = (double)(h->box_hi[0] - h->box_lo[0]) * (h->box_hi[1] - h->box_lo[1]) *
                      (h->box_hi[2] - h->box_lo[2]);
        out->density = h->total_mass / out->volume;
        // W = sum r_i . F_i: pair + 1-4 + bonds + excluded-pair corrections + reciprocal sum + constraints;
        // the uniform neutralising background of a charged cell scales as 1/V (W = 3 E), the Ewald self term not at all
        out->virial = e[EN_VIRIAL] + (h->pme_on ? 3.0 * h->ewald_background : 0.0);
        out->pressure = (2.0 * out->kinetic + out->virial) / (3.0 * out->volume) * MDX_BAR_PER_KCAL_MOL_A3;
    }
    if (h->alch_on) {
        // (with the SPME reciprocal sum: + its dU/dlambda = -2 E_env,mol; the real-space cross energy is the "coupled interaction")
        out->dh_dlambda = du_dl + (h->pme_on ? e[EN_COUNT + 5] : 0.0);
        out->coupled_interaction = (1.0 - h->alch_lambda) * (u_cross - (h->pme_on ? e[EN_COUNT + 5] : 0.0));
    }
    uint32_t mf2; std::memcpy(&mf2, &e[EN_COUNT], 4);
    out->max_force = std::sqrt((double)u2f(mf2));
    const double tot = out->potential + out->kinetic;
    if (!std::isfinite(tot)) FAIL(MDX_ENAN, "non-finite energy");
    return MDX_OK;
}

// The stateless scorer is called pose after pose on the same molecules (src/docking/mod.rs:235): building and tearing
// down the device state costs 8 ms at 50 k atoms, scoring a pose 0.15 ms.  The calling thread therefore keeps the handle
// of its last call; a call whose system has the same STATIC content (everything but coordinates, velocities and box:
// compared by a 64-bit fingerprint of the arrays) and the same config only uploads the new coordinates.
// (mdx_hostutil.cpp: plain C++ - the AVX2 flavour needs <immintrin.h>, which a HIP translation unit cannot include)
uint64_t mdx_fp_mix(uint64_t h, const void* p, size_t bytes);
static inline uint64_t fp_mix(uint64_t h, const void* p, size_t bytes) { return mdx_fp_mix(h, p, bytes); }
static uint64_t system_fingerprint(const mdx_system* s, const mdx_config* c, int device) {
    uint64_t h = 0x243F6A8885A308D3ull ^ (uint64_t)(uint32_t)device;
    const size_t N = s->n_atoms;
    const uint32_t head[12] = {s->n_atoms, s->n_lj_types, s->n_bonds, s->n_angles, s->n_dihedrals, s->n_pairs14, s->n_mols,
                               (uint32_t)s->periodic, s->n_constraints, s->n_vsites, 0u, 0u};
    h = fp_mix(h, head, sizeof(head));
    h = fp_mix(h, c, sizeof(*c));
    h = fp_mix(h, s->mass, 4 * N); h = fp_mix(h, s->charge, 4 * N); h = fp_mix(h, s->lj_type, 4 * N);
    h = fp_mix(h, s->lj_sigma, 4 * (size_t)s->n_lj_types); h = fp_mix(h, s->lj_eps, 4 * (size_t)s->n_lj_types);
    h = fp_mix(h, s->flags, s->flags ? N : 0);
    h = fp_mix(h, s->bond_idx, 8 * (size_t)s->n_bonds); h = fp_mix(h, s->bond_k, 4 * (size_t)s->n_bonds); h = fp_mix(h, s->bond_r0, 4 * (size_t)s->n_bonds);
    h = fp_mix(h, s->angle_idx, 12 * (size_t)s->n_angles); h = fp_mix(h, s->angle_k, 4 * (size_t)s->n_angles);
    h = fp_mix(h, s->angle_theta0, 4 * (size_t)s->n_angles);
    h = fp_mix(h, s->dihedral_idx, 16 * (size_t)s->n_dihedrals); h = fp_mix(h, s->dihedral_v, 4 * (size_t)s->n_dihedrals);
    h = fp_mix(h, s->dihedral_phase, 4 * (size_t)s->n_dihedrals); h = fp_mix(h, s->dihedral_n, 4 * (size_t)s->n_dihedrals);
    if (s->excl_offsets) { h = fp_mix(h, s->excl_offsets, 4 * (N + 1)); h = fp_mix(h, s->excl_idx, 4 * (size_t)s->excl_offsets[N]); }
    h = fp_mix(h, s->pairs14_idx, 8 * (size_t)s->n_pairs14);
    h = fp_mix(h, s->mol_start, s->mol_start ? 4 * (size_t)s->n_mols : 0);
    h = fp_mix(h, s->constraint_idx, 8 * (size_t)s->n_constraints); h = fp_mix(h, s->constraint_len, 4 * (size_t)s->n_constraints);
    h = fp_mix(h, s->vsite_idx, 16 * (size_t)s->n_vsites); h = fp_mix(h, s->vsite_w, 8 * (size_t)s->n_vsites);
    return h ? h : 1;
}
struct SinglePointCache {
    mdx_handle* h = nullptr; uint64_t key = 0;
    int device = 0; mdx_config cfg{}; uint32_t counts[10]{};      // what the optimistic path compares before it trusts the handle
    static void counts_of(const mdx_system* s, uint32_t* c) {
        const uint32_t v[10] = {s->n_atoms, s->n_lj_types, s->n_bonds, s->n_angles, s->n_dihedrals, s->n_pairs14, s->n_mols,
                                (uint32_t)s->periodic, s->n_constraints, s->n_vsites};
        std::memcpy(c, v, sizeof(v));
    }
    void remember_counts(const mdx_system* s) { counts_of(s, counts); }
    bool counts_match(const mdx_system* s) const { uint32_t c[10]; counts_of(s, c); return std::memcmp(c, counts, sizeof(c)) == 0; }
    std::vector<float> pos, vel;      // coordinates / velocities of the last pose (vel empty = none given)
    // A worker thread that scored poses and exits must not leak its handle (device buffers, stream, FFT plans).  At
    // process exit the HIP runtime may already be gone: only destroy when it still answers.
    ~SinglePointCache() {
        int n = 0;
        if (h && hipGetDeviceCount(&n) == hipSuccess && n > 0) mdx_destroy(h);
        h = nullptr;
    }
};
static thread_local SinglePointCache g_sp_cache;

extern "C" int mdx_single_point_between_mols(const mdx_system* sys, const mdx_config* cfg, int device, const uint8_t* group_of_atom,
                                             uint32_t n_groups, mdx_energies* out, float* forces_or_null, float* matrix_out) {
    if (!matrix_out) FAIL(MDX_EPARAM, "null matrix_out");
    // matrix_out holds n_groups^2 floats: with the by-molecule map that must be the system's molecule count (a caller that sized the
    // buffer by another number would be overrun)
    if (sys && !group_of_atom && n_groups != sys->n_mols)
        FAIL(MDX_EPARAM, "group_of_atom == NULL groups by molecule: n_groups must equal mdx_system.n_mols (the matrix is n_mols x n_mols)");
    MDX_TRY(mdx_single_point(sys, cfg, device, out, forces_or_null));
    mdx_handle* h = g_sp_cache.h;      // the pose just scored lives on the calling thread's kept handle
    if (!h) FAIL(MDX_EDEVICE, "internal: the scorer kept no handle");
    // the group map is part of the request, not of the system: set it when it differs from what the kept handle holds
    const uint32_t N = h->N;
    bool same = h->n_grp != 0;
    if (group_of_atom) same = same && h->n_grp == n_groups && h->grp_host.size() == N && std::memcmp(h->grp_host.data(), group_of_atom, N) == 0 && !h->grp_by_mol;
    else same = same && h->grp_by_mol;
    if (!same) MDX_TRY(mdx_set_energy_groups(h, group_of_atom, group_of_atom ? n_groups : 0u));
    return mdx_groups_evaluate(h, matrix_out);
}

extern "C" void mdx_single_point_release(void) {
    if (g_sp_cache.h) { std::string keep = g_last_error; mdx_destroy(g_sp_cache.h); g_last_error = keep; }
    g_sp_cache.h = nullptr; g_sp_cache.key = 0;
    g_sp_cache.pos.clear(); g_sp_cache.vel.clear();
}

extern "C" int mdx_single_point(const mdx_system* sys, const mdx_config* cfg, int device, mdx_energies* out,
                                float* forces_or_null) {
    // A pose of the molecules the calling thread scored last (same static content, by fingerprint) was validated when its
    // device state was built: only the coordinates that changed are checked again (mdx_upload_range).  Anything else
    // goes through the full validation.
    if (!sys || !cfg) FAIL(MDX_EPARAM, "null system or config");
    if (sys->n_atoms == 0 || !sys->pos || !sys->mass || !sys->charge || !sys->lj_type || !sys->lj_sigma || !sys->lj_eps)
        MDX_TRY(validate(sys, cfg));
    mdx_handle* h = nullptr;
    // Docking moves the ligand, ~50 atoms of ~50 k (src/docking/mod.rs:81-154): only the span of atoms whose
    // coordinates differ from the last pose travels, and while they stay within skin/2 of where the Verlet list
    // was built the list is reused (mdx_upload_range) - a pose then costs one energy-flavoured force pass.
    const size_t n3 = 3 * (size_t)sys->n_atoms;
    auto changed_span = [&](const float* now, const std::vector<float>& last, uint32_t* first, uint32_t* count) {
        size_t a = 0, b = n3;
        const size_t blk = 1024;   // whole blocks by memcmp (vectorised), then word by word inside the first / last differing block
        while (a + blk <= n3 && std::memcmp(&now[a], &last[a], blk * sizeof(float)) == 0) a += blk;
        while (a < n3 && std::memcmp(&now[a], &last[a], sizeof(float)) == 0) ++a;
        if (a == n3) { *first = 0; *count = 0; return; }
        while (b >= a + blk && std::memcmp(&now[b - blk], &last[b - blk], blk * sizeof(float)) == 0) b -= blk;
        while (b > a && std::memcmp(&now[b - 1], &last[b - 1], sizeof(float)) == 0) --b;
        *first = (uint32_t)(a / 3); *count = (uint32_t)((b + 2) / 3) - *first;
    };
    auto new_pose = [&](mdx_handle* hh) -> int {       // same molecules, new pose
        int rc = MDX_OK;
        if (sys->periodic) {
            bool same = true;
            for (int d = 0; d < 3; ++d) same &= hh->box_lo[d] == sys->box_lo[d] && hh->box_hi[d] == sys->box_hi[d];
            if (!same) rc = mdx_set_box(hh, sys->box_lo, sys->box_hi);
        }
        uint32_t f0 = 0, cnt = 0;
        if (rc == MDX_OK) {
            changed_span(sys->pos, g_sp_cache.pos, &f0, &cnt);
            rc = mdx_upload_range(hh, MDX_POS, f0, cnt, sys->pos + 3 * (size_t)f0);
            if (rc == MDX_OK && cnt) std::memcpy(&g_sp_cache.pos[3 * (size_t)f0], sys->pos + 3 * (size_t)f0, sizeof(float) * 3 * cnt);
        }
        if (rc == MDX_OK) {
            if (sys->vel) {
                if (g_sp_cache.vel.empty()) { f0 = 0; cnt = sys->n_atoms; g_sp_cache.vel.assign(sys->vel, sys->vel + n3); }
                else {
                    changed_span(sys->vel, g_sp_cache.vel, &f0, &cnt);
                    if (cnt) std::memcpy(&g_sp_cache.vel[3 * (size_t)f0], sys->vel + 3 * (size_t)f0, sizeof(float) * 3 * cnt);
                }
                rc = mdx_upload_range(hh, MDX_VEL, f0, cnt, sys->vel + 3 * (size_t)f0);
            } else if (!g_sp_cache.vel.empty()) {
                std::vector<float> z(n3, 0.f);
                rc = mdx_upload(hh, MDX_VEL, z.data());
                g_sp_cache.vel.clear();
            }
        }
        return rc;
    };
    // A pose of the molecules this thread scored last was validated when its device state was built.  Whether the call IS such a
    // pose is decided by a 64-bit fingerprint of the static arrays (~0.1 ms at 50 k atoms) - computed while the device evaluates the
    // pose on the cached handle (the evaluation is 0.15 ms), not in front of it: a call that passes the cheap comparisons (atom and
    // term counts, config) uploads its changed coordinates and enqueues the evaluation optimistically; if the fingerprint then
    // disagrees - or anything on the way fails - the result is dropped and the call goes the long way (validate, build, evaluate).
    struct FpJob { const mdx_system* s; const mdx_config* c; int device; uint64_t key; };
    if (g_sp_cache.h && g_sp_cache.h->N == sys->n_atoms && g_sp_cache.device == device && std::memcmp(&g_sp_cache.cfg, cfg, sizeof(*cfg)) == 0 &&
        g_sp_cache.counts_match(sys) && out) {
        h = g_sp_cache.h;
        FpJob job{sys, cfg, device, 0};
        static const bool dbg = [] { const char* e = std::getenv("MDX_DEBUG_SP"); return e && e[0] == '1'; }();
        const auto t0 = std::chrono::steady_clock::now();
        int rc = new_pose(h);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == MDX_OK) {
            h->wait_hook_arg = &job;
            h->wait_hook = [](void* p) { FpJob* j = (FpJob*)p; j->key = system_fingerprint(j->s, j->c, j->device); };
            rc = mdx_energy(h, out);
            if (h->wait_hook) { h->wait_hook = nullptr; job.key = system_fingerprint(sys, cfg, device); }    // (served from a cache: no wait happened)
        }
        if (dbg) {
            const auto t2 = std::chrono::steady_clock::now();
            static double a = 0, b = 0; static int n = 0;
            a += std::chrono::duration<double, std::micro>(t1 - t0).count(); b += std::chrono::duration<double, std::micro>(t2 - t1).count();
            if (++n % 20 == 0) std::fprintf(stderr, "[mdx single_point] pose update %.1f us, evaluation (fingerprint inside) %.1f us\n", a / n, b / n);
        }
        if (rc == MDX_OK && job.key == g_sp_cache.key) {
            if (forces_or_null) rc = mdx_download(h, MDX_FORCE, forces_or_null);
            if (rc != MDX_OK) mdx_single_point_release();
            return rc;
        }
        mdx_single_point_release();      // other molecules after all (or an error a full validation will name): the long way
        h = nullptr;
    }
    MDX_TRY(validate(sys, cfg));
    {
        const uint64_t key = system_fingerprint(sys, cfg, device);
        if (g_sp_cache.h && g_sp_cache.key == key && g_sp_cache.h->N == sys->n_atoms) {
            h = g_sp_cache.h;
            const int rc = new_pose(h);
            if (rc != MDX_OK) { mdx_single_point_release(); return rc; }
        } else {
            mdx_single_point_release();
            MDX_TRY(mdx_create(sys, cfg, device, &h));
            g_sp_cache.h = h; g_sp_cache.key = key; g_sp_cache.device = device; g_sp_cache.cfg = *cfg; g_sp_cache.remember_counts(sys);
            g_sp_cache.pos.assign(sys->pos, sys->pos + 3 * (size_t)sys->n_atoms);
            if (sys->vel) g_sp_cache.vel.assign(sys->vel, sys->vel + 3 * (size_t)sys->n_atoms);
        }
    }
    int rc = out ? mdx_energy(h, out) : MDX_OK;
    if (rc == MDX_OK && forces_or_null) rc = mdx_download(h, MDX_FORCE, forces_or_null);
    if (rc != MDX_OK) mdx_single_point_release();   // never keep a handle in an unknown state
    return rc;
}

// ---------------------------------------------------------------------------------------------
extern "C" int mdx_download(mdx_handle* h, int which, float* dst) {
    if (!h || !dst) FAIL(MDX_EPARAM, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->dd) {   // decomposed: a collective - every rank receives the global array (all ranks must make the call)
        if (which != MDX_POS && which != MDX_VEL && which != MDX_FORCE) FAIL(MDX_EPARAM, "unknown array selector");
        return mdx_dd_download(h, which, dst);
    }
    hipStream_t st = h->stream;
    const uint32_t N = h->n_local;   // == n_atoms unless mdx_set_local_atoms narrowed the set
    const float4* src = nullptr;
    if (which == MDX_POS || which == MDX_VEL) {
        if (h->in_slot_space)
            MDX_TRY(mdx_gather_to_orig(h, which == MDX_POS ? h->d.posq : h->d.vel,
                                       which == MDX_POS ? h->d.pos_orig : h->d.vel_orig));
        src = which == MDX_POS ? h->d.pos_orig : h->d.vel_orig;
    } else if (which == MDX_FORCE) {
        MDX_TRY(ensure_ready(h));
        // caller-order scratch kept with the handle (the scorer reads forces pose after pose: no malloc per call)
        if (!h->d.scratch4 || h->cap_scratch4 < N) {
            if (h->d.scratch4) { (void)hipFree(h->d.scratch4); h->d.scratch4 = nullptr; }
            HIP_TRY(hipMalloc((void**)&h->d.scratch4, sizeof(float4) * (size_t)std::max(N, h->cap_local)));
            h->cap_scratch4 = std::max(N, h->cap_local);
        }
        MDX_TRY(mdx_gather_to_orig(h, h->d.force, h->d.scratch4));
        std::vector<float4> hf(N);
        HIP_TRY(hipMemcpyAsync(hf.data(), h->d.scratch4, sizeof(float4) * N, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint32_t i = 0; i < N; ++i) {
            const bool ghost = h->n_local == h->N && (h->flags[i] & MDX_ATOM_GHOST) != 0;
            dst[3 * i] = ghost ? 0.f : hf[i].x; dst[3 * i + 1] = ghost ? 0.f : hf[i].y;
            dst[3 * i + 2] = ghost ? 0.f : hf[i].z;
        }
        return MDX_OK;
    } else FAIL(MDX_EPARAM, "unknown array selector");
    std::vector<float4> hb(N);
    HIP_TRY(hipMemcpyAsync(hb.data(), src, sizeof(float4) * N, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < N; ++i) { dst[3 * i] = hb[i].x; dst[3 * i + 1] = hb[i].y; dst[3 * i + 2] = hb[i].z; }
    return MDX_OK;
}

extern "C" int mdx_upload(mdx_handle* h, int which, const float* src) {
    if (!h || !src) FAIL(MDX_EPARAM, "null argument");
    h->e_cache_valid = false; h->e_pending = false;
    if (which != MDX_POS && which != MDX_VEL) FAIL(MDX_EPARAM, "only MDX_POS and MDX_VEL can be uploaded");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->dd && h->n_local != h->N) FAIL(MDX_EPARAM, "mdx_upload on a handle narrowed by mdx_set_local_atoms (the host that drives the decomposition owns the state)");
    const uint32_t N = h->N;
    for (size_t k = 0; k < 3 * (size_t)N; ++k)
        if (!std::isfinite(src[k])) FAIL(MDX_EPARAM, "non-finite value in upload");
    std::vector<float4> b(N);
    for (uint32_t i = 0; i < N; ++i) {
        b[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.f);
        if (which == MDX_VEL && (h->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST))) b[i] = make_float4(0, 0, 0, 0);
    }
    if (h->dd) {      // joined handle: collective, every rank passes the same array (mdx_decomp.hip)
        MDX_TRY(mdx_dd_upload(h, which, 0, N, b.data()));
        if (mdx_has_constraints(h)) h->cons_dirty = true;
        return MDX_OK;
    }
    MDX_TRY(mdx_unsort_state(h));  // keep the other array: both now live in caller-order staging
    HIP_TRY(hipMemcpyAsync(which == MDX_POS ? h->d.pos_orig : h->d.vel_orig, b.data(), sizeof(float4) * N,
                           hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->list_valid = false; h->forces_valid = false;
    if (mdx_has_constraints(h)) h->cons_dirty = true;
    return MDX_OK;
}

// New coordinates for atoms [first, first + count) while the spatial caches stay: each atom is put at the periodic image
// nearest to where it was at the last list build, and the largest displacement from there decides (as the drift pass
// does in the step loop) whether the Verlet list still covers the new pose.
__global__ __launch_bounds__(256) void pose_update_kernel(uint32_t first, uint32_t count, const float4* __restrict__ src,
                                                          const uint32_t* __restrict__ slot_of, float4* __restrict__ posq,
                                                          const float4* __restrict__ ref, float lx, float ly, float lz,
                                                          uint32_t* __restrict__ flag, uint32_t thr_bits) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    float d2 = 0.f;
    if (k < count) {
        const uint32_t s = slot_of[first + k];
        if (s != MDX_INVALID) {
            const float4 p = src[k], r = ref[s];
            float dx = p.x - r.x, dy = p.y - r.y, dz = p.z - r.z;
            if (lx > 0.f) dx -= rintf(dx / lx) * lx;
            if (ly > 0.f) dy -= rintf(dy / ly) * ly;
            if (lz > 0.f) dz -= rintf(dz / lz) * lz;
            float4 q = posq[s];
            q.x = r.x + dx; q.y = r.y + dy; q.z = r.z + dz;
            posq[s] = q;
            d2 = dx * dx + dy * dy + dz * dz;
            if (!(d2 < 1.0e30f)) d2 = 3.0e38f;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(d2) > thr_bits) atomicMax(flag, __float_as_uint(d2));
}

extern "C" int mdx_upload_range(mdx_handle* h, int which, uint32_t first, uint32_t count, const float* src) {
    if (!h || (count && !src)) FAIL(MDX_EPARAM, "null argument");
    h->e_cache_valid = false; h->e_pending = false;
    if (which != MDX_POS && which != MDX_VEL) FAIL(MDX_EPARAM, "only MDX_POS and MDX_VEL can be uploaded");
    if (!h->dd && h->n_local != h->N) FAIL(MDX_EPARAM, "mdx_upload_range on a handle narrowed by mdx_set_local_atoms");
    if ((uint64_t)first + count > h->N) FAIL(MDX_EPARAM, "atom range out of bounds");
    if (count == 0) return MDX_OK;
    HIP_TRY(hipSetDevice(h->device));
    for (size_t k = 0; k < 3 * (size_t)count; ++k)
        if (!std::isfinite(src[k])) FAIL(MDX_EPARAM, "non-finite value in upload");
    hipStream_t st = h->stream;
    std::vector<float4> b(count);
    for (uint32_t k = 0; k < count; ++k) {
        b[k] = make_float4(src[3 * k], src[3 * k + 1], src[3 * k + 2], 0.f);
        if (which == MDX_VEL && (h->flags[first + k] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST))) b[k] = make_float4(0, 0, 0, 0);
    }
    if (h->dd) {      // joined handle: collective (the moved rows may change owner: gather, overwrite, repartition)
        MDX_TRY(mdx_dd_upload(h, which, first, count, b.data()));
        if (mdx_has_constraints(h)) h->cons_dirty = true;
        return MDX_OK;
    }
    const bool keep_list = which == MDX_POS && h->in_slot_space && h->list_valid && !std::isinf(h->r_list);
    if (!keep_list) {
        MDX_TRY(mdx_unsort_state(h));
        HIP_TRY(hipMemcpyAsync((which == MDX_POS ? h->d.pos_orig : h->d.vel_orig) + first, b.data(), sizeof(float4) * count,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        h->list_valid = false; h->forces_valid = false;
        if (mdx_has_constraints(h)) h->cons_dirty = true;
        return MDX_OK;
    }
    // pose update in slot space: ext_orig is free between steps (mdx_step re-installs it), use its head as the stage
    if (h->have_ext) FAIL(MDX_EPARAM, "mdx_upload_range while external forces are installed");
    HIP_TRY(hipMemcpyAsync(h->d.ext_orig + first, b.data(), sizeof(float4) * count, hipMemcpyHostToDevice, st));
    const uint32_t thr = stale_threshold_bits(h);
    HIP_TRY(hipMemsetAsync(&h->d.ctl->disp2[0], 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(pose_update_kernel, dim3(div_up(count, 256)), dim3(256), 0, st, first, count, h->d.ext_orig + first,
                       h->d.slot_of, h->d.posq, h->d.ref, h->per[0] ? h->box_hi[0] - h->box_lo[0] : 0.f,
                       h->per[1] ? h->box_hi[1] - h->box_lo[1] : 0.f, h->per[2] ? h->box_hi[2] - h->box_lo[2] : 0.f,
                       &h->d.ctl->disp2[0], thr);
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, &h->d.ctl->disp2[0], sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(&h->d.ctl->disp2[0], 0, sizeof(uint32_t), st));
    HIP_TRY(hipStreamSynchronize(st));
    h->forces_valid = false;
    h->prune_pending = true; h->moved_outside = true;   // dual list: the path accumulators did not see this move
    if (flag > thr) h->list_valid = false; // moved further than skin/2 from the list's reference: rebuild on next use
    if (mdx_has_constraints(h)) h->cons_dirty = true;
    return MDX_OK;
}

extern "C" int mdx_set_box(mdx_handle* h, const float lo[3], const float hi[3]) {
    if (!h || !lo || !hi) FAIL(MDX_EPARAM, "null argument");
    if (!h->periodic && !h->dd) FAIL(MDX_EPARAM, "mdx_set_box on a non-periodic system");
    {   // (a rank of a fully cut box is not periodic locally: the box is still the global, fully periodic one)
        const int per_all[3] = {1, 1, 1};
        MDX_TRY(check_box(h->dd ? per_all : h->per, lo, hi, &h->cfg));
    }
    HIP_TRY(hipSetDevice(h->device));
    if (h->dd) { h->e_cache_valid = false; h->e_pending = false; return mdx_dd_set_box(h, lo, hi, nullptr, nullptr); }
    MDX_TRY(mdx_unsort_state(h));
    for (int d = 0; d < 3; ++d) { h->box_lo[d] = lo[d]; h->box_hi[d] = hi[d]; }
    h->list_valid = false; h->forces_valid = false;
    MDX_TRY(mdx_pme_setup(h));     // mesh spacing and theta(m) follow the box
    return MDX_OK;
}

extern "C" int mdx_rebuild_spatial_caches(mdx_handle* h) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    h->list_valid = false;
    MDX_TRY(mdx_rebuild(h));
    return MDX_OK;
}

extern "C" int mdx_neighbor_list(mdx_handle* h, uint32_t* offsets, uint32_t* idx) {
    if (!h || !offsets) FAIL(MDX_EPARAM, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->list_valid) MDX_TRY(mdx_rebuild(h));
    return mdx_extract_neighbors(h, offsets, idx);
}

extern "C" int mdx_profile(mdx_handle* h, int enable) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    mdx_prof_collect(h);
    h->profile = enable != 0; h->profile_level = enable;
    if (enable) {
        h->stats.nb_ms_sum = h->stats.bonded_ms_sum = h->stats.integ_ms_sum = 0.0;
        h->stats.rebuild_ms_sum = h->stats.wall_ms_sum = 0.0;
        h->stats.nb_launches = h->stats.bonded_launches = h->stats.integ_launches = 0;
        if (h->dd) for (int k = 0; k < MDX_DIAG_PHASES; ++k) { h->dd->phase_ms[k] = 0.0; h->dd->phase_n[k] = 0; }
    }
    return MDX_OK;
}

extern "C" int mdx_get_skin(const mdx_handle* h, float* skin, int* tuning) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    if (skin) *skin = h->cfg.skin;
    if (tuning) *tuning = (h->skin_tune.on && h->skin_tune.phase < 4) ? 1 : 0;
    return MDX_OK;
}

extern "C" int mdx_pair_launch_info(const mdx_handle* h, uint32_t out[24]) {
    if (!h || !out) FAIL(MDX_EPARAM, "null argument");
    for (int k = 0; k < 8; ++k) { out[k] = h->pair_info_step[k]; out[8 + k] = h->pair_info_any[k]; }
    out[16] = h->inner_rebuilds; out[17] = h->inner_from_rebuild ? 1u : 0u; out[18] = h->water_step_launches; out[19] = h->water_step_mixed_launches;
    out[20] = (uint32_t)h->onepass_launches; out[21] = (uint32_t)h->onepass_violations; out[22] = 0u; out[23] = 0u;
    return MDX_OK;
}

extern "C" int mdx_get_stats(mdx_handle* h, mdx_stats* out) {
    if (!h || !out) FAIL(MDX_EPARAM, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    mdx_prof_collect(h);
    h->stats.step_count = h->step_count; h->stats.rebuild_count = h->rebuild_count;
    h->stats.n_inner_cluster_pairs = 0; h->stats.prune_passes = 0;
    if (h->d.inner_count) {   // dual list: cumulative kept cluster pairs (spread over MDX_EPART words) and passes
        unsigned long long c[MDX_EPART + 1];
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(c, h->d.inner_count, sizeof(c), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        unsigned long long tot = 0;
        for (int k = 0; k < MDX_EPART; ++k) tot += c[k];
        h->stats.prune_passes = c[MDX_EPART];
        h->stats.n_inner_cluster_pairs = c[MDX_EPART] ? tot / c[MDX_EPART] : 0;   // mean per pass
    }
    h->stats.n_owned = h->dd ? h->dd->n_owned : h->n_local;
    h->stats.n_ghost = h->dd ? h->dd->n_local - h->dd->n_owned : 0;
    h->stats.repartitions = h->dd ? h->dd->repartitions : 0;
    h->stats.local_rebuilds = h->dd ? h->dd->local_rebuilds : 0;
    h->stats.repartition_ms_sum = h->dd ? h->dd->repartition_ms : 0.0;
    *out = h->stats;
    return MDX_OK;
}

// ---- multi-GPU support: split step + halo pack/unpack -------------------------------------------
// A row whose id is MDX_INVALID is a FLAG row: it carries this rank's rebuild-flag word to the peer
// (bit pattern in .x), so the global "list went stale" decision rides on the halo message itself.
__global__ void pack_pos_kernel(uint32_t n, const uint32_t* __restrict__ atom_idx, const uint32_t* __restrict__ slot_of,
                                const float4* __restrict__ posq, float4* __restrict__ out,
                                const uint32_t* __restrict__ flag_word) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = atom_idx[i];
    if (g == MDX_INVALID) {
        out[i] = make_float4(flag_word ? __uint_as_float(*flag_word) : 0.f, 0.f, 0.f, 0.f);
        return;
    }
    const uint32_t s = slot_of[g];
    out[i] = (s == MDX_INVALID) ? make_float4(0.f, 0.f, 0.f, 0.f) : posq[s];
}
// Dual pair list: a ghost moves by halo message, not by the drift pass, so its path accumulator (ref[].w) is fed here.
__global__ void unpack_pos_kernel(uint32_t n, const uint32_t* __restrict__ atom_idx, const uint32_t* __restrict__ slot_of,
                                  float4* __restrict__ posq, const float4* __restrict__ in,
                                  const float4* __restrict__ shift, uint32_t* __restrict__ flag_word,
                                  float4* __restrict__ ref, uint32_t* __restrict__ prune_out, float path_thr,
                                  float gx, float gy, float gz, float* __restrict__ path_arr) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = atom_idx[i];
    if (g == MDX_INVALID) {   // a peer's flag word: merge (max) into ours
        if (flag_word) atomicMax(flag_word, __float_as_uint(in[i].x));
        return;
    }
    const uint32_t s = slot_of[g];
    if (s == MDX_INVALID) return;
    float4 v = in[i];
    if (shift) { const float4 sh = shift[i]; v.x += sh.x; v.y += sh.y; v.z += sh.z; }
    float4 p = posq[s];
    // (library-managed decomposition: gx, gy, gz = the global box) The ghost goes to the periodic image of the incoming
    // position that is nearest to where it stands now.  The message carries the SENDER's coordinates; for an atom the
    // sender owns through its cluster's anchor while it lies across the periodic face (a straddling rigid water) those
    // are a box length away from the wrapped coordinate the receiver's image shift was computed for, and an atom that
    // has drifted out of [lo, lo + L) on its owner is in the same position.  A ghost moves by far less than L / 2 per
    // message and starts in the right image (the partition places it), so the nearest image is the true one.
    if (gx > 0.f) v.x -= gx * rintf((v.x - p.x) / gx);
    if (gy > 0.f) v.y -= gy * rintf((v.y - p.y) / gy);
    if (gz > 0.f) v.z -= gz * rintf((v.z - p.z) / gz);
    if (ref) {
        const float mx = v.x - p.x, my = v.y - p.y, mz = v.z - p.z;
        const float mv = sqrtf(mx * mx + my * my + mz * mz);
        float w;
        if (path_arr) { w = path_arr[s] + mv; path_arr[s] = w; } else { w = ref[s].w + mv; ref[s].w = w; }
        if (!(w <= path_thr)) *prune_out = 1u;   // idempotent, rare
    }
    p.x = v.x; p.y = v.y; p.z = v.z;
    posq[s] = p;
}

extern "C" int mdx_set_local_atoms(mdx_handle* h, uint32_t n_local, const uint32_t* d_gid, const uint8_t* d_ghost,
                                   const float* d_pos4, const float* d_vel4, const float lo[3], const float hi[3],
                                   int32_t periodic) {
    if (!h || !d_gid || !d_ghost || !d_pos4 || !d_vel4 || !lo || !hi) FAIL(MDX_EPARAM, "null argument");
    if (h->dd) FAIL(MDX_EPARAM, "the handle's decomposition is managed by the library (mdx_comm_init)");
    if (mdx_has_constraints(h) || h->n_vsites)
        FAIL(MDX_EPARAM, "constraints / virtual sites need cluster-wise ownership: use mdx_comm_init, which decides it");
    return mdx_set_local_atoms_impl(h, n_local, d_gid, d_ghost, d_pos4, d_vel4, lo, hi, periodic);
}

int mdx_set_local_atoms_impl(mdx_handle* h, uint32_t n_local, const uint32_t* d_gid, const uint8_t* d_ghost,
                             const float* d_pos4, const float* d_vel4, const float lo[3], const float hi[3],
                             int32_t periodic) {
    if (n_local == 0 || n_local > h->N) FAIL(MDX_EPARAM, "n_local must be in 1..n_atoms (each atom at most once)");
    if (h->pme_on && !h->dd) FAIL(MDX_EPARAM, "the SPME reciprocal sum needs the library's own decomposition (mdx_comm_init): here each rank would spread only its own charges");
    if (h->alch_on && !h->dd) FAIL(MDX_EPARAM, "alchemical windows need the library's own decomposition (mdx_comm_init)");
    if (h->integrator != MDX_INTEGRATOR_VERLET_VELOCITY && !h->dd) FAIL(MDX_EPARAM, "only velocity Verlet is supported when the host drives the decomposition itself (mdx_comm_init handles every integrator)");
    if (h->baro_kind && !h->dd) FAIL(MDX_EPARAM, "the barostat needs the library's own decomposition (mdx_comm_init)");
    HIP_TRY(hipSetDevice(h->device));
    int per[3];
    decode_periodic(periodic, per);
    MDX_TRY(check_box(per, h->box_lo, h->box_hi, &h->cfg));
    hipStream_t st = h->stream;
    DeviceState& d = h->d;
    // (the library's own decomposition fills these arrays in place: nothing to copy then)
    if (d_gid != d.gid) HIP_TRY(hipMemcpyAsync(d.gid, d_gid, sizeof(uint32_t) * n_local, hipMemcpyDeviceToDevice, st));
    if (d_ghost != d.lflag) HIP_TRY(hipMemcpyAsync(d.lflag, d_ghost, sizeof(uint8_t) * n_local, hipMemcpyDeviceToDevice, st));
    if ((const void*)d_pos4 != (const void*)d.pos_orig) HIP_TRY(hipMemcpyAsync(d.pos_orig, d_pos4, sizeof(float4) * n_local, hipMemcpyDeviceToDevice, st));
    if ((const void*)d_vel4 != (const void*)d.vel_orig) HIP_TRY(hipMemcpyAsync(d.vel_orig, d_vel4, sizeof(float4) * n_local, hipMemcpyDeviceToDevice, st));
    h->n_local = n_local;
    for (int k = 0; k < 3; ++k) { h->per[k] = per[k]; h->local_lo[k] = lo[k]; h->local_hi[k] = hi[k]; }
    h->periodic = per[0] || per[1] || per[2];
    h->have_local_bounds = true;
    h->in_slot_space = false; h->list_valid = false; h->forces_valid = false;
    h->slot_of_clean = false;      // atoms that left the local set still have their old slots in d.slot_of
    return MDX_OK;
}

extern "C" int mdx_local_state(mdx_handle* h, float* d_pos4, float* d_vel4) {
    if (!h || !d_pos4 || !d_vel4) FAIL(MDX_EPARAM, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->in_slot_space) {
        MDX_TRY(mdx_gather_to_orig(h, h->d.posq, (float4*)d_pos4));
        MDX_TRY(mdx_gather_to_orig(h, h->d.vel, (float4*)d_vel4));
    } else {
        HIP_TRY(hipMemcpyAsync(d_pos4, h->d.pos_orig, sizeof(float4) * h->n_local, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(d_vel4, h->d.vel_orig, sizeof(float4) * h->n_local, hipMemcpyDeviceToDevice, h->stream));
    }
    return MDX_OK;
}

extern "C" int mdx_chunk_begin(mdx_handle* h) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->list_valid) MDX_TRY(mdx_rebuild(h));
    HIP_TRY(hipMemsetAsync(h->d.ctl, 0, sizeof(StepCtl), h->stream));
    h->chunk_s = -1;
    return MDX_OK;
}

extern "C" int mdx_chunk_integrate(mdx_handle* h, int mode, float dt, uint32_t s) {
    if (!h || s > MDX_MAX_CHUNK || mode < 0 || mode > 2) FAIL(MDX_EPARAM, "bad argument");
    h->prof_tag = (int)s;
    if (mode != 2) h->chunk_s = (int)s;   // the halo unpack and the force call of this step share its prune word
    int rc = mdx_launch_integrate(h, mode, dt, &h->d.ctl->disp2[s], mode == 2 ? nullptr : &h->d.ctl->disp2[s + 1],
                                  stale_threshold_bits(h), mode == 2 ? nullptr : &h->d.ctl->prune[s + 1]);
    h->prof_tag = -1;
    return rc;
}

extern "C" int mdx_chunk_forces(mdx_handle* h, int32_t s) {
    if (!h || s > (int32_t)MDX_MAX_CHUNK) FAIL(MDX_EPARAM, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->list_valid) MDX_TRY(mdx_rebuild(h));
    h->prof_tag = s;
    h->nb_step = s;   // s >= 0: a step-loop force call, may walk the inner list / prune on ctl.prune[s + 1]
    int rc = compute_forces(h, false, s < 0 ? nullptr : &h->d.ctl->disp2[s + 1], stale_threshold_bits(h));
    h->nb_step = -1;
    h->prof_tag = -1;
    if (rc == MDX_OK) h->forces_valid = true;
    return rc;
}

extern "C" int mdx_chunk_end(mdx_handle* h, uint32_t n_words, uint32_t* flags_out) {
    if (!h || !flags_out || n_words > MDX_MAX_CHUNK + 2) FAIL(MDX_EPARAM, "bad argument");
    HIP_TRY(hipMemcpyAsync(h->h_ctl, h->d.ctl, sizeof(StepCtl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const uint32_t thr = stale_threshold_bits(h);
    int first_stale = 1 << 30;
    for (uint32_t k = 0; k < n_words; ++k) {
        flags_out[k] = h->h_ctl->disp2[k];
        if (k > 0 && first_stale == (1 << 30) && flags_out[k] > thr) first_stale = (int)k - 1;
    }
    if (h->profile) mdx_prof_collect(h, first_stale);
    return MDX_OK;
}

extern "C" void* mdx_flag_words(mdx_handle* h) { return h ? (void*)h->d.ctl->disp2 : nullptr; }
extern "C" uint32_t mdx_stale_threshold(const mdx_handle* h) { return h ? stale_threshold_bits(h) : 0u; }
extern "C" int mdx_add_steps(mdx_handle* h, uint32_t n) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    h->step_count += n;
    return MDX_OK;
}

extern "C" int mdx_pack_positions(mdx_handle* h, const uint32_t* d_gid, uint32_t n, float* d_out4, int32_t flag_word) {
    if (!h || (n && (!d_gid || !d_out4)) || flag_word > (int32_t)MDX_MAX_CHUNK + 1) FAIL(MDX_EPARAM, "bad argument");
    if (!h->in_slot_space) FAIL(MDX_EPARAM, "spatial caches not built");
    if (n) hipLaunchKernelGGL(pack_pos_kernel, dim3(div_up(n, 256)), dim3(256), 0, h->stream, n, d_gid,
                              h->d.slot_of, h->d.posq, (float4*)d_out4,
                              flag_word >= 0 ? &h->d.ctl->disp2[flag_word] : nullptr);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

extern "C" int mdx_unpack_positions(mdx_handle* h, const uint32_t* d_gid, uint32_t n, const float* d_in4,
                                    const float* d_shift4, int32_t flag_word) {
    if (!h || (n && (!d_gid || !d_in4)) || flag_word > (int32_t)MDX_MAX_CHUNK + 1) FAIL(MDX_EPARAM, "bad argument");
    if (!h->in_slot_space) FAIL(MDX_EPARAM, "spatial caches not built");
    const bool dual = h->dual_on && h->chunk_s >= 0;
    if (n) hipLaunchKernelGGL(unpack_pos_kernel, dim3(div_up(n, 256)), dim3(256), 0, h->stream, n, d_gid,
                              h->d.slot_of, h->d.posq, (const float4*)d_in4, (const float4*)d_shift4,
                              flag_word >= 0 ? &h->d.ctl->disp2[flag_word] : nullptr,
                              dual ? h->d.ref : nullptr,
                              dual ? (mdx_dd_split_now(h) ? &h->d.ctl->prune_ghost[h->chunk_s + 1]
                                                                                                      : &h->d.ctl->prune[h->chunk_s + 1]) : nullptr,
                              0.5f * h->inner_skin * (1.0f - 1.0e-4f),
                              h->dd ? h->box_hi[0] - h->box_lo[0] : 0.f, h->dd ? h->box_hi[1] - h->box_lo[1] : 0.f,
                              h->dd ? h->box_hi[2] - h->box_lo[2] : 0.f, (dual && h->path_split) ? h->d.path : nullptr);
    HIP_TRY(hipGetLastError());
    h->forces_valid = false;
    if (!dual) h->moved_outside = true;     // ghosts moved without feeding their path accumulators
    return MDX_OK;
}

extern "C" void* mdx_stream(mdx_handle* h) { return h ? (void*)h->stream : nullptr; }
extern "C" int mdx_debug_half_stats(mdx_handle* h, unsigned long long out[6]) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->d.inner_count + MDX_EPART, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost));
    return MDX_OK;
}
