// mdx_bonded.hip — Amber bonded terms (bond / angle / dihedral) and scaled 1-4 pairs.
//
// Energy forms per the reference's parameter records, /root/reference src/ui/popup/ff_params.rs:
//   bond     E = k_b (r - r_0)^2                         :352-372
//   angle    E = k (theta - theta_0)^2                   :401-421
//   dihedral E = (barrier_height/divider) [1 + cos(periodicity*phi - phase)]   :474-511
//
// Atom-owned gather, no atomics: one lane per slot walks the atom's own list of "roles" (one
// 16-byte record per (term, atom-of-that-term): the partner slots, the term kind, the atom's
// position in the term, the index of its parameter set), recomputes each term it belongs to and keeps only the
// force on itself.  A bond is evaluated twice, an angle three times, a dihedral four times — the
// arithmetic is free next to the memory traffic — and in exchange every force component has a
// single writer: f[slot] += sum is a plain read-modify-write, the result is bitwise reproducible,
// and the 7 M contended f32 atomics per step of a term-per-lane scatter (0.35 ms on the 1 M-atom
// water box, 20x the streaming time) are gone.  Role lists live in slot space and are rebuilt at
// every neighbour rebuild, so a lane's records are contiguous and its partners sit in the same
// or an adjacent tile (L1/L2 hits).  HBM-bound: ~36 + 16 r B per atom-step, r = roles per atom
// (2.33 for water).
#include "mdx_bonded_dev.h"

struct BondedArgs {
    uint32_t S;
    const uint32_t* role_off; const RoleRec* roles; const float4* prm;
    const float4* posq; float4* force; double* energy;
    BondedParams p;
    const uint32_t* gate; uint32_t thr_bits;
};

// Four lanes per atom: lane q of the quad takes roles q, q+4, ... of the atom's list and the quad sums its
// forces with two DPP steps.  A chain atom has ~20 roles (each a chain of dependent partner loads) against a
// water atom's 2.3; with one lane per atom every wave waited for its longest list (23 k-atom solvated chain:
// 26 us, as long as the pair kernel) - and a quad reads four consecutive 32-B role records as one 128-B line.
template <int CTRL>
__device__ __forceinline__ float quad_xadd(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// BONDED_LPA lanes per atom: 4 when the system has chains (above), 2 for mixed and 1 for pure-solvent systems - a flexible
// water atom has 2.3 roles, so with four lanes per atom most of them only read the offsets and leave (1 M-atom water:
// 36.6 us with 4 lanes, 30.7 with 2, 27.8 with 1).
template <bool ENERGY, int BONDED_LPA>
__global__ __launch_bounds__(256) void bonded_gather_kernel(BondedArgs a) {
    if (a.gate && *a.gate > a.thr_bits) return;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = tid / BONDED_LPA, q4 = tid % BONDED_LPA;
    RoleEnergies en;
    if (s < a.S) {
        const uint32_t rb = a.role_off[s], re = a.role_off[s + 1];
        if (re > rb) {     // quad-uniform
            const float4 self = a.posq[s];
            float fx = 0.f, fy = 0.f, fz = 0.f;
            for (uint32_t k = rb + q4; k < re; k += BONDED_LPA) {
                const RoleRec r = a.roles[k];
                role_eval<ENERGY>(r, a.prm, self, a.posq, a.p, fx, fy, fz, en);
            }
            if (BONDED_LPA >= 2) { fx = quad_xadd<0xB1>(fx); fy = quad_xadd<0xB1>(fy); fz = quad_xadd<0xB1>(fz); }   // lane ^ 1
            if (BONDED_LPA == 4) { fx = quad_xadd<0x4E>(fx); fy = quad_xadd<0x4E>(fy); fz = quad_xadd<0x4E>(fz); }   // lane ^ 2
            if (q4 == 0) {
                float4 f = a.force[s];
                f.x += fx; f.y += fy; f.z += fz;
                a.force[s] = f;
            }
        }
    }
    if (ENERGY) {
        // wave shuffle, then the block's four waves through LDS: one atomic per term and block (one per wave
        // was 7 x 16 k contended atomics at 1 M atoms: 0.42 ms for a 0.04 ms kernel)
        __shared__ double s_e[4][7];
        double v[7] = {en.bond, en.angle, en.dih, en.lj14, en.c14, en.rec, en.vir};
        const int slot[7] = {EN_BOND, EN_ANGLE, EN_DIHEDRAL, EN_LJ14, EN_COUL14, EN_RECIP, EN_VIRIAL};
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            double t = v[q];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m);
            if ((threadIdx.x & 63) == 0) s_e[threadIdx.x >> 6][q] = t;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const double t = s_e[0][threadIdx.x] + s_e[1][threadIdx.x] + s_e[2][threadIdx.x] + s_e[3][threadIdx.x];
            if (t != 0.0) atomicAdd(&a.energy[slot[threadIdx.x]], t);
        }
    }
}

// `ext` is in the caller's (global) atom order; `gid` maps a local atom to it (identity on a whole handle).  On a decomposed
// handle every rank is given the same global array and adds the rows of the atoms it owns (a ghost's force is never used).
__global__ void add_ext_kernel(uint32_t S, const uint32_t* __restrict__ orig_of, const uint32_t* __restrict__ gid,
                               const uint8_t* __restrict__ slot_flags, const float4* __restrict__ ext,
                               float4* __restrict__ force, const uint32_t* gate, uint32_t thr_bits) {
    if (gate && *gate > thr_bits) return;
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    uint32_t o = orig_of[s];
    if (o == MDX_INVALID || !(slot_flags[s] & 2u)) return;
    float4 f = force[s], e = ext[gid[o]];
    f.x += e.x; f.y += e.y; f.z += e.z;
    force[s] = f;
}

bool mdx_bonded_wanted(const mdx_handle* h) {
    const bool skip_bonded = (h->cfg.overrides & MDX_OVR_BONDED_DISABLED) != 0;
    return h->n_roles && !(skip_bonded && !h->pme_on);
}
void mdx_fill_bonded_params(const mdx_handle* h, BondedParams& p) {
    for (int d = 0; d < 3; ++d) {
        p.box[d] = h->per[d] ? (h->box_hi[d] - h->box_lo[d]) : 0.f;
        p.inv_box[d] = h->per[d] ? 1.0f / p.box[d] : 0.f;
    }
    p.ewald_beta = h->cfg.ewald_alpha;
    p.skip_bonded = (h->cfg.overrides & MDX_OVR_BONDED_DISABLED) ? 1 : 0;
}

int mdx_launch_bonded(mdx_handle* h, bool energy, const uint32_t* d_gate, uint32_t thr_bits) {
    if (h->bonded_fused) { h->bonded_fused = false; return MDX_OK; }   // rode along with the pair launch (small systems)
    if (h->bonded_deferred && !energy) return MDX_OK;                  // the next step's fused bonded + kick + drift pass evaluates them
    if (!mdx_bonded_wanted(h)) return MDX_OK;
    BondedArgs a{};
    a.S = h->S; a.role_off = h->d.role_off_s; a.roles = h->d.role_rec_s; a.prm = h->d.role_prm;
    a.posq = h->d.posq; a.force = h->d.force; a.energy = h->d.energy; a.gate = d_gate; a.thr_bits = thr_bits;
    mdx_fill_bonded_params(h, a.p);
    mdx_prof_begin(h, energy ? 3 : 1);
    // lanes per atom by the mean role count of the slots (MDX_BONDED_LPA=2|4 forces it for A/B)
    static const int lpa_env = [] { const char* e = std::getenv("MDX_BONDED_LPA"); return e ? std::atoi(e) : 0; }();
    // (n_roles counts the whole system: a decomposed handle holds n_local of its N atoms, and only the owned ones carry roles)
    const double roles_here = (double)h->n_roles * ((h->n_local != h->N && h->N) ? (double)h->n_local / (double)h->N : 1.0);
    const int lpa = (lpa_env == 1 || lpa_env == 2 || lpa_env == 4) ? lpa_env : (roles_here < 2.6 * (double)h->S ? 1 : (roles_here < 6.0 * (double)h->S ? 2 : 4));
    const dim3 g((uint32_t)(((size_t)h->S * lpa + 255) / 256)), b(256);
    if (lpa == 1) {
        if (energy) hipLaunchKernelGGL((bonded_gather_kernel<true, 1>), g, b, 0, h->stream, a);
        else hipLaunchKernelGGL((bonded_gather_kernel<false, 1>), g, b, 0, h->stream, a);
    } else if (lpa == 2) {
        if (energy) hipLaunchKernelGGL((bonded_gather_kernel<true, 2>), g, b, 0, h->stream, a);
        else hipLaunchKernelGGL((bonded_gather_kernel<false, 2>), g, b, 0, h->stream, a);
    } else {
        if (energy) hipLaunchKernelGGL((bonded_gather_kernel<true, 4>), g, b, 0, h->stream, a);
        else hipLaunchKernelGGL((bonded_gather_kernel<false, 4>), g, b, 0, h->stream, a);
    }
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_add_ext(mdx_handle* h, const uint32_t* d_gate, uint32_t thr_bits) {
    if (!h->have_ext) return MDX_OK;
    hipLaunchKernelGGL(add_ext_kernel, dim3((h->S + 255) / 256), dim3(256), 0, h->stream, h->S, h->d.orig_of, h->d.gid,
                       h->d.slot_flags, h->d.ext_orig, h->d.force, d_gate, thr_bits);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}
