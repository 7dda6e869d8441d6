// mdx_groups.hip - `SnapshotEnergyData.energy_potential_between_mols`: the non-bonded energy between molecules (or caller-chosen
// groups of atoms) as an n x n matrix.
//
// Consumers in the reference: /root/reference src/properties/crystal.rs:347-370 (`cohesive_energy_from_matrix`: flat row-major
// n_mol x n_mol, the upper triangle is summed) called at :533 on every snapshot's `energy_potential_between_mols`;
// src/ui/panels/md_viewer.rs:234-237 shows element [1].  It is also what a docking scorer wants of BASELINE config 3
// (src/docking/mod.rs:81-154): the receptor - ligand interaction energy instead of the total potential.  The arithmetic lives in
// the absent `dynamics` crate; built here:
//
//   M[a][b] = M[b][a] = sum over atom pairs (i in group a, j in group b) of the pair loop's LJ + Coulomb energy (the configured
//             real-space Coulomb treatment: shifted cutoff / reaction field / erfc(beta r)/r, same cutoffs, exclusions and images as
//             the forces) + the scaled 1-4 energy of 1-4 pairs between the two groups;
//   M[a][a] = the same sums over pairs inside group a (each pair once).
//   => sum over a <= b of M[a][b] = mdx_energies.lj + coulomb + lj14 + coulomb14  (= potential_nonbonded - coulomb_recip: the SPME
//      mesh part is a property of the whole charge density and is not split by group).
//
// One dedicated pass over the Verlet list (the pair kernels of the step loop are not touched): the lane mapping of
// nb_cluster_kernel - lane = (i-atom ii of every i-cluster, j-atom jj of the entry), j-clusters staged through the wave's LDS
// strip - calling the SAME pair_eval (mdx_pair_dev.h) in its energy flavour, so a matrix element is made of exactly the pair
// energies mdx_energy sums.  A lane keeps one running fp64 sum and the (group_i, group_j) key it belongs to; the sum leaves
// through an fp64 atomic when the key changes (molecules are spatially compact: in a solvated complex nearly every lane of
// nearly every tile sees one key) and at the end of the tile, where lanes with equal keys are combined first.  Rare work
// (snapshots, docking poses): correctness and bit-for-bit the same pair terms first; ~1.3 x the energy flavour's own time.
#include "mdx_bonded_dev.h"
#include "mdx_comm.h"
#include "mdx_pair_dev.h"
#include <cstring>

#define FAIL(code, msg) do { mdx_set_error(msg); return (code); } while (0)

struct GroupArgs {
    NbArgs nb;
    const uint32_t* orig_of; const uint32_t* gid; const uint8_t* grp;   // slot -> local atom -> global atom -> group
    uint32_t G; double* mat;                                            // [G * G] raw sums, row = group of the i-atom
    uint32_t half;                                                      // the list holds every cluster pair once
    uint32_t mask_layout;                                               // exclusion masks: 2 = bit (8 e + ci) of lane (ii, jj) (the cluster kernels); 1 = bit (8 e + jj) of lane i-atom (nb_variant 1)
};

__device__ __forceinline__ uint32_t grp_of_slot(const GroupArgs& a, uint32_t s) {
    const uint32_t o = a.orig_of[s];
    return o == MDX_INVALID ? 0u : (uint32_t)a.grp[a.gid[o]];
}

template <int COUL, bool GEOM, bool ALCH>
__global__ __launch_bounds__(256) void nb_group_kernel(GroupArgs ga) {
    const NbArgs& a = ga.nb;
    __shared__ float4 s_xyzq[4][64];
    __shared__ float2 s_lj[4][64];
    __shared__ uint32_t s_meta[4][64];      // group of the staged j-atom | owned << 8
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t t = blockIdx.x;          // one workgroup = one tile, wave w takes chunks w, w + 4, ...
    if (t >= a.T) return;
    const int ii = lane & 7, jj = lane >> 3;
    float xi[8], yi[8], zi[8], qi[8], sgi[8], epi[8];
    uint32_t gi_lo = 0, gi_hi = 0, own_bits = 0;
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const uint32_t s = t * MDX_TILE + ci * MDX_CLUSTER + ii;
        const float4 pi = a.posq[s];
        const float2 li = a.lj[s];
        xi[ci] = pi.x; yi[ci] = pi.y; zi[ci] = pi.z; qi[ci] = pi.w; sgi[ci] = li.x; epi[ci] = li.y;
        const uint32_t g = grp_of_slot(ga, s);
        if (ci < 4) gi_lo |= g << (8 * ci); else gi_hi |= g << (8 * (ci - 4));
        own_bits |= (a.energy_all ? 1u : ((a.slot_flags[s] >> 1) & 1u)) << ci;
    }
    const ListCounts cnt = a.counts[t];
    const uint32_t e0 = a.entry_off[t], nmc = cnt.n_masked >> 3, nchunks = (cnt.n_masked + cnt.n_plain) >> 3;
    const uint32_t mbase = a.mchunk_off[t];
    uint32_t cur_key = 0xFFFFFFFFu;
    double dacc = 0.0;
    for (uint32_t c = (uint32_t)wave; c < nchunks; c += 4) {
        const uint2 ent = a.entries[e0 + c * 8 + (lane >> 3)];
        const uint32_t js = ent.x * MDX_CLUSTER + (lane & 7);
        float4 nj = a.posq[js];
        {
            const uint32_t code = ent.y & 31u;
            const int kx = (int)(code % 3u) - 1, ky = (int)((code / 3u) % 3u) - 1, kz = (int)(code / 9u) - 1;
            nj.x += (float)kx * a.p.shift[0]; nj.y += (float)ky * a.p.shift[1]; nj.z += (float)kz * a.p.shift[2];
        }
        unsigned long long mq = c < nmc ? a.masks[(size_t)(mbase + c) * 64 + lane] : ~0ull;
        if (ga.mask_layout == 1 && c < nmc) {      // the whole-tile kernel's layout: transpose it into this lane mapping (as the list build does for layout 2)
            const unsigned long long m = mq;
            mq = 0ull;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const unsigned long long mp = __shfl(m, ci * 8 + ii);
#pragma unroll
                for (int e = 0; e < 8; ++e) mq |= ((mp >> (8 * e + jj)) & 1ull) << (8 * e + ci);
            }
        }
        s_xyzq[wave][lane] = nj;
        s_lj[wave][lane] = a.lj[js];
        s_meta[wave][lane] = grp_of_slot(ga, js) | ((a.energy_all ? 1u : ((a.slot_flags[js] >> 1) & 1u)) << 8);
        WAVE_LDS_SYNC();
#pragma unroll 1
        for (int e = 0; e < 8; ++e) {
            const uint32_t im = (__builtin_amdgcn_readlane(ent.y, e * 8) >> 8) & 0xFFu;     // wave-uniform
            if (im == 0) continue;
            const float4 pj = s_xyzq[wave][e * 8 + jj];
            const float2 lj = s_lj[wave][e * 8 + jj];
            const uint32_t mj = s_meta[wave][e * 8 + jj];
            const uint32_t gj = mj & 0xFFu, own_j = (mj >> 8) & 1u;
            const uint32_t allowed8 = (uint32_t)(mq >> (8 * e)) & 0xFFu;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (!(im & (1u << ci))) continue;
                float fx = 0.f, fy = 0.f, fz = 0.f, e1 = 0.f, e2 = 0.f;
                const float bias = ((allowed8 >> ci) & 1u) ? 0.f : __builtin_nanf("");
                pair_eval<true, COUL, GEOM, false, true, false, ALCH, true>(xi[ci], yi[ci], zi[ci], qi[ci], sgi[ci], epi[ci], pj, lj, true, a.p,
                                                                         fx, fy, fz, e1, e2, nullptr, nullptr, nullptr, bias);
                const float es = e1 + e2;
                if (es == 0.f) continue;
                const uint32_t own_i = (own_bits >> ci) & 1u;
                // half list: the pair is evaluated once on this rank (and, without the half shell, once on the rank that owns the other
                // atom: half weight per owned atom); full list: every pair comes by twice, once from each side
                const float w = ga.half ? 0.5f * (float)(own_i + own_j) : (own_i ? 0.5f : 0.f);
                const uint32_t g_i = ((ci < 4 ? gi_lo >> (8 * ci) : gi_hi >> (8 * (ci - 4)))) & 0xFFu;
                const uint32_t key = g_i * ga.G + gj;
                if (key != cur_key) {
                    if (dacc != 0.0) atomicAdd(ga.mat + cur_key, dacc);
                    cur_key = key; dacc = 0.0;
                }
                dacc += (double)(w * es);
            }
        }
        WAVE_LDS_SYNC();
    }
    // lanes with equal keys leave as one atomic
    unsigned long long pending = __ballot(dacc != 0.0);
    while (pending) {
        const int src = __ffsll((long long)pending) - 1;
        const uint32_t k = (uint32_t)__shfl((int)cur_key, src);
        const bool mine = dacc != 0.0 && cur_key == k;
        double v = mine ? dacc : 0.0;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) atomicAdd(ga.mat + k, v);
        pending &= ~__ballot(mine);
    }
}

// the scaled 1-4 pairs: credited once, by the atom in role 0 (as the bonded gather does for EN_LJ14 / EN_COUL14); roles exist on owned
// atoms only, so every pair is counted on exactly one rank
__global__ __launch_bounds__(256) void group_pairs14_kernel(uint32_t S, const uint32_t* __restrict__ role_off, const RoleRec* __restrict__ roles,
                                                            const float4* __restrict__ prm, const float4* __restrict__ posq, BondedParams p,
                                                            GroupArgs ga) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S || p.skip_bonded) return;
    const uint32_t rb = role_off[s], re = role_off[s + 1];
    if (re <= rb) return;
    const float4 self = posq[s];
    const uint32_t g_s = grp_of_slot(ga, s);
    for (uint32_t k = rb; k < re; ++k) {
        const RoleRec r = roles[k];
        if ((r.meta & 0xFu) != ROLE_PAIR14 || ((r.meta >> 4) & 0xFu) != 0u) continue;
        const float4 pr = prm[r.meta >> 8];      // sigma_ij, 4 scale eps_ij, scale k_e q_i q_j
        const float3 d = mimg(sub3(self, posq[r.p[0]]), p);
        const float r2 = dot3(d, d), rinv = rsqrtf(r2), rinv2 = rinv * rinv;
        const float s2 = pr.x * pr.x * rinv2, s6 = s2 * s2 * s2;
        const double e = (double)(pr.y * s6 * (s6 - 1.0f)) + (double)(pr.z * rinv);
        if (e != 0.0) atomicAdd(ga.mat + g_s * ga.G + grp_of_slot(ga, r.p[0]), e);
    }
}

template <int COUL>
static void launch_groups(mdx_handle* h, const GroupArgs& ga, bool geom) {
    const dim3 g(ga.nb.T), b(256);
    if (h->alch_on) {
        if (geom) hipLaunchKernelGGL((nb_group_kernel<COUL, true, true>), g, b, 0, h->stream, ga);
        else hipLaunchKernelGGL((nb_group_kernel<COUL, false, true>), g, b, 0, h->stream, ga);
    } else {
        if (geom) hipLaunchKernelGGL((nb_group_kernel<COUL, true, false>), g, b, 0, h->stream, ga);
        else hipLaunchKernelGGL((nb_group_kernel<COUL, false, false>), g, b, 0, h->stream, ga);
    }
}

// The matrix of the current state (symmetric, row-major, n = the number of groups), evaluated afresh.  Collective on a decomposed handle.
int mdx_groups_evaluate(mdx_handle* h, float* out) {
    const uint32_t G = h->n_grp;
    if (!G) FAIL(MDX_EPARAM, "no energy groups are set (mdx_set_energy_groups)");
    MDX_TRY(mdx_ensure_ready(h));          // list, constraints, ghosts and virtual sites of the current state
    hipStream_t st = h->stream;
    DeviceState& d = h->d;
    GroupArgs ga{};
    NbArgs& a = ga.nb;
    a.T = h->T; a.posq = d.posq; a.lj = d.lj; a.counts = d.list_counts; a.entry_off = d.entry_off; a.mchunk_off = d.mchunk_off;
    a.entries = d.entries; a.masks = d.masks; a.slot_flags = d.slot_flags;
    a.energy_all = mdx_dd_half_shell(h) ? 1u : 0u;
    int mode = 0; bool geom = false, samecut = false;
    mdx_fill_nb_params(h, a.p, &mode, &geom, &samecut);
    ga.orig_of = d.orig_of; ga.gid = d.gid; ga.grp = d.grp; ga.G = G; ga.mat = d.grp_mat; ga.half = mdx_nb_half(h) ? 1u : 0u;
    ga.mask_layout = mdx_nb_variant(h) >= 2 ? 2u : 1u;
    HIP_TRY(hipMemsetAsync(d.grp_mat, 0, sizeof(double) * (size_t)G * G, st));
    if (h->T) {
        switch (mode) {
        case CM_SHIFTED: launch_groups<CM_SHIFTED>(h, ga, geom); break;
        case CM_SOFT: launch_groups<CM_SOFT>(h, ga, geom); break;
        case CM_RF: launch_groups<CM_RF>(h, ga, geom); break;
        default: launch_groups<CM_EWALD>(h, ga, geom); break;
        }
    }
    if (h->n_p14 && h->n_roles) {
        BondedParams bp{};
        mdx_fill_bonded_params(h, bp);
        hipLaunchKernelGGL(group_pairs14_kernel, dim3((h->S + 255) / 256), dim3(256), 0, st, h->S, d.role_off_s, d.role_rec_s, d.role_prm, d.posq, bp, ga);
    }
    HIP_TRY(hipGetLastError());
    if (h->dd && h->dd->world > 1) MDX_TRY(mdx_dd_allreduce_dev(h, d.grp_mat, (size_t)G * G));
    std::vector<double> raw((size_t)G * G);
    HIP_TRY(hipMemcpyAsync(raw.data(), d.grp_mat, sizeof(double) * raw.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < G; ++i)
        for (uint32_t j = 0; j < G; ++j) {
            const double v = i == j ? raw[(size_t)i * G + i] : raw[(size_t)i * G + j] + raw[(size_t)j * G + i];
            if (!std::isfinite(v)) FAIL(MDX_ENAN, "non-finite energy between groups");
            out[(size_t)i * G + j] = (float)v;
        }
    return MDX_OK;
}

extern "C" int mdx_set_energy_groups(mdx_handle* h, const uint8_t* group_of_atom, uint32_t n_groups) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t N = h->N;
    std::vector<uint8_t> g(N, 0);
    uint32_t G = n_groups;
    if (!group_of_atom) {
        if (n_groups == MDX_GROUPS_OFF || (n_groups == 0 && h->mol_start.empty())) {      // off (explicitly, or as before on a system without molecules)
            if (h->d.grp) { (void)hipFree(h->d.grp); h->d.grp = nullptr; }
            if (h->d.grp_mat) { (void)hipFree(h->d.grp_mat); h->d.grp_mat = nullptr; }
            h->n_grp = 0; h->grp_host.clear();
            return MDX_OK;
        }
        if (n_groups != 0) FAIL(MDX_EPARAM, "a NULL group map takes n_groups = 0 (one group per molecule) or MDX_GROUPS_OFF");
        // by molecule: md.mol_start_indices (src/md/mod.rs:809-947)
        if (h->mol_start.empty()) FAIL(MDX_EPARAM, "grouping by molecule needs mol_start in the system description");
        if (h->mol_start.size() > MDX_MAX_GROUPS) FAIL(MDX_EPARAM, "more than 255 molecules: pass a group map (mdx_set_energy_groups) of at most 255 groups");
        G = (uint32_t)h->mol_start.size();
        for (uint32_t m = 0; m < G; ++m) {
            const uint32_t lo = h->mol_start[m], hi = m + 1 < G ? h->mol_start[m + 1] : N;
            if (lo > hi || hi > N) FAIL(MDX_EPARAM, "mol_start must ascend within the atom range");
            for (uint32_t i = lo; i < hi; ++i) g[i] = (uint8_t)m;
        }
        if (h->mol_start[0] != 0) FAIL(MDX_EPARAM, "mol_start[0] must be 0");
    } else {
        if (G == 0 || G > MDX_MAX_GROUPS) FAIL(MDX_EPARAM, "n_groups must be in 1..255");
        for (uint32_t i = 0; i < N; ++i) {
            if (group_of_atom[i] >= G) FAIL(MDX_EPARAM, "group index out of range");
            g[i] = group_of_atom[i];
        }
    }
    if (h->d.grp) { (void)hipFree(h->d.grp); h->d.grp = nullptr; }
    if (h->d.grp_mat) { (void)hipFree(h->d.grp_mat); h->d.grp_mat = nullptr; }
    HIP_TRY(hipMalloc((void**)&h->d.grp, std::max<size_t>(N, 16)));
    HIP_TRY(hipMalloc((void**)&h->d.grp_mat, sizeof(double) * (size_t)G * G));
    HIP_TRY(hipMemcpyAsync(h->d.grp, g.data(), N, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->grp_host.swap(g); h->n_grp = G; h->grp_by_mol = group_of_atom == nullptr;
    return MDX_OK;
}

extern "C" uint32_t mdx_energy_group_count(const mdx_handle* h) { return h ? h->n_grp : 0u; }

extern "C" int mdx_energy_between_mols(mdx_handle* h, float* out, uint32_t n) {
    if (!h || !out) FAIL(MDX_EPARAM, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->n_grp) FAIL(MDX_EPARAM, "no energy groups are set (mdx_set_energy_groups)");
    if (n != h->n_grp) FAIL(MDX_EPARAM, "n must be the number of groups (mdx_energy_group_count)");
    return mdx_groups_evaluate(h, out);
}
