// mdx_bonded.hip — Amber bonded terms (bond / angle / dihedral) and scaled 1-4 pairs.
//
// Energy forms per the reference's parameter records, /root/reference src/ui/popup/ff_params.rs:
//   bond     E = k_b (r - r_0)^2                         :352-372
//   angle    E = k (theta - theta_0)^2                   :401-421
//   dihedral E = (barrier_height/divider) [1 + cos(periodicity*phi - phase)]   :474-511
//
// Atom-owned gather, no atomics: one lane per slot walks the atom's own list of "roles" (one
// 32-byte record per (term, atom-of-that-term): the partner slots, the term kind, the atom's
// position in the term, the parameters), recomputes each term it belongs to and keeps only the
// force on itself.  A bond is evaluated twice, an angle three times, a dihedral four times — the
// arithmetic is free next to the memory traffic — and in exchange every force component has a
// single writer: f[slot] += sum is a plain read-modify-write, the result is bitwise reproducible,
// and the 7 M contended f32 atomics per step of a term-per-lane scatter (0.35 ms on the 1 M-atom
// water box, 20x the streaming time) are gone.  Role lists live in slot space and are rebuilt at
// every neighbour rebuild, so a lane's records are contiguous and its partners sit in the same
// or an adjacent tile (L1/L2 hits).  HBM-bound: ~36 + 32 r B per atom-step, r = roles per atom
// (2.33 for water).
#include "mdx_internal.h"

struct BondedArgs {
    uint32_t S;
    const uint32_t* role_off; const RoleRec* roles;
    const float4* posq; float4* force; double* energy;
    BondedParams p;
    const uint32_t* gate; uint32_t thr_bits;
};

__device__ __forceinline__ float3 mimg(float3 d, const BondedParams& p) {
    // minimum image d - rint(d/L) L (src/cuda/util.cu:65-71); box[] = 0 disables it in vacuum
    if (p.box[0] > 0.f) d.x -= rintf(d.x * p.inv_box[0]) * p.box[0];
    if (p.box[1] > 0.f) d.y -= rintf(d.y * p.inv_box[1]) * p.box[1];
    if (p.box[2] > 0.f) d.z -= rintf(d.z * p.inv_box[2]) * p.box[2];
    return d;
}
__device__ __forceinline__ float3 sub3(float4 a, float4 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float3 scale3(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }

// Four lanes per atom: lane q of the quad takes roles q, q+4, ... of the atom's list and the quad sums its
// forces with two DPP steps.  A chain atom has ~20 roles (each a chain of dependent partner loads) against a
// water atom's 2.3; with one lane per atom every wave waited for its longest list (23 k-atom solvated chain:
// 26 us, as long as the pair kernel) - and a quad reads four consecutive 32-B role records as one 128-B line.
template <int CTRL>
__device__ __forceinline__ float quad_xadd(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int BONDED_LPA = 4;

template <bool ENERGY>
__global__ __launch_bounds__(256) void bonded_gather_kernel(BondedArgs a) {
    if (a.gate && *a.gate > a.thr_bits) return;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = tid / BONDED_LPA, q4 = tid % BONDED_LPA;
    double e_bond = 0.0, e_angle = 0.0, e_dih = 0.0, e_lj14 = 0.0, e_c14 = 0.0, e_rec = 0.0, e_vir = 0.0;
    if (s < a.S) {
        const uint32_t rb = a.role_off[s], re = a.role_off[s + 1];
        if (re > rb) {     // quad-uniform
            const float4 self = a.posq[s];
            float fx = 0.f, fy = 0.f, fz = 0.f;
            for (uint32_t k = rb + q4; k < re; k += BONDED_LPA) {
                const RoleRec r = a.roles[k];
                const uint32_t kind = r.meta & 0xFu, role = (r.meta >> 4) & 0xFu;
                if (kind == ROLE_EWALD_EXCL) {
                    // the reciprocal sum includes this excluded / 1-4 pair: take erf(beta r)/r out again
                    const float3 d = mimg(sub3(self, a.posq[r.p[0]]), a.p);
                    const float r2 = dot3(d, d), rinv = rsqrtf(r2), rr = r2 * rinv, br = a.p.ewald_beta * rr;
                    const float er = erff(br);
                    const float fs = -r.prm[0] * (er * rinv - 1.1283791671f * a.p.ewald_beta * __expf(-br * br)) * rinv * rinv;
                    fx += fs * d.x; fy += fs * d.y; fz += fs * d.z;
                    if (ENERGY && role == 0) { e_rec -= (double)(r.prm[0] * er * rinv); e_vir += (double)(fs * r2); }
                    continue;
                }
                if (a.p.skip_bonded) continue;
                if (kind == ROLE_BOND || kind == ROLE_PAIR14) {
                    const float3 d = mimg(sub3(self, a.posq[r.p[0]]), a.p);
                    const float r2 = dot3(d, d);
                    float fs;
                    if (kind == ROLE_BOND) {
                        const float rr = sqrtf(r2), dr = rr - r.prm[1];
                        fs = -2.0f * r.prm[0] * dr / rr;
                        if (ENERGY && role == 0) e_bond += (double)r.prm[0] * dr * dr;
                    } else {   // prm: sigma_ij, 4*scale*eps_ij, scale*ke*qi*qj
                        const float rinv = rsqrtf(r2), rinv2 = rinv * rinv;
                        const float s2 = r.prm[0] * r.prm[0] * rinv2, s6 = s2 * s2 * s2;
                        fs = (6.0f * r.prm[1] * s6 * (2.0f * s6 - 1.0f) + r.prm[2] * rinv) * rinv2;
                        if (ENERGY && role == 0) {
                            e_lj14 += (double)(r.prm[1] * s6 * (s6 - 1.0f));
                            e_c14 += (double)(r.prm[2] * rinv);
                        }
                    }
                    fx += fs * d.x; fy += fs * d.y; fz += fs * d.z;
                    // virial: central pair forces only - angle and dihedral energies do not change
                    // under a uniform scaling, their sum r_i . F_i is identically zero
                    if (ENERGY && role == 0) e_vir += (double)(fs * r2);
                } else if (kind == ROLE_ANGLE) {
                    // ordered atoms i - j(apex) - k; this lane is atom `role`
                    const float4 q0 = a.posq[r.p[0]], q1 = a.posq[r.p[1]];
                    const float4 pi = role == 0 ? self : q0;
                    const float4 pj = role == 1 ? self : (role == 0 ? q0 : q1);
                    const float4 pk = role == 2 ? self : q1;
                    const float3 v1 = mimg(sub3(pi, pj), a.p), v2 = mimg(sub3(pk, pj), a.p);
                    const float ir1 = rsqrtf(dot3(v1, v1)), ir2 = rsqrtf(dot3(v2, v2));
                    float cs = dot3(v1, v2) * ir1 * ir2;
                    cs = fminf(1.0f, fmaxf(-1.0f, cs));
                    const float th = acosf(cs), dth = th - r.prm[1];
                    const float sn = fmaxf(sqrtf(1.0f - cs * cs), 1e-6f);
                    const float de = 2.0f * r.prm[0] * dth;   // dE/dtheta
                    // dtheta/dr_i = -(v2/|v2| - cos v1/|v1|) / (|v1| sin)
                    const float ci = de * ir1 / sn, ck = de * ir2 / sn;
                    const float3 u1 = scale3(v1, ir1), u2 = scale3(v2, ir2);
                    const float3 fi = make_float3(ci * (u2.x - cs * u1.x), ci * (u2.y - cs * u1.y), ci * (u2.z - cs * u1.z));
                    const float3 fk = make_float3(ck * (u1.x - cs * u2.x), ck * (u1.y - cs * u2.y), ck * (u1.z - cs * u2.z));
                    if (role == 0) { fx += fi.x; fy += fi.y; fz += fi.z; }
                    else if (role == 2) { fx += fk.x; fy += fk.y; fz += fk.z; }
                    else { fx -= fi.x + fk.x; fy -= fi.y + fk.y; fz -= fi.z + fk.z; }
                    if (ENERGY && role == 0) e_angle += (double)r.prm[0] * dth * dth;
                } else {   // ROLE_DIHEDRAL: ordered atoms 0-1-2-3, this lane is atom `role`
                    const float4 q0 = a.posq[r.p[0]], q1 = a.posq[r.p[1]], q2 = a.posq[r.p[2]];
                    const float4 p0 = role == 0 ? self : q0;
                    const float4 p1 = role == 1 ? self : (role == 0 ? q0 : q1);
                    const float4 p2 = role == 2 ? self : (role < 2 ? q1 : q2);
                    const float4 p3 = role == 3 ? self : q2;
                    // Blondel-Karplus: F = r0-r1, G = r1-r2, H = r3-r2, A = F x G, B = H x G
                    const float3 F = mimg(sub3(p0, p1), a.p), G = mimg(sub3(p1, p2), a.p), H = mimg(sub3(p3, p2), a.p);
                    const float3 A = cross3(F, G), B = cross3(H, G);
                    const float A2 = dot3(A, A), B2 = dot3(B, B), G2 = dot3(G, G);
                    if (A2 > 1e-12f && B2 > 1e-12f && G2 > 1e-12f) {
                        const float Gn = sqrtf(G2), iGn = 1.0f / Gn;
                        const float cosphi = dot3(A, B);
                        const float sinphi = dot3(cross3(B, A), G) * iGn;
                        const float phi = atan2f(sinphi, cosphi);
                        float sn, cn;
                        sincosf(r.prm[2] * phi - r.prm[1], &sn, &cn);
                        const float de = -r.prm[0] * r.prm[2] * sn;   // dE/dphi
                        const float iA2 = 1.0f / A2, iB2 = 1.0f / B2;
                        const float FG = dot3(F, G), HG = dot3(H, G);
                        const float ca = -Gn * iA2, cb = Gn * iB2;
                        const float ta = FG * iA2 * iGn, tb = HG * iB2 * iGn;
                        // f_x = -dE/dphi * dphi/dr_x = -de * (wa * A + wb * B)
                        float wa, wb;
                        if (role == 0) { wa = ca; wb = 0.f; }
                        else if (role == 3) { wa = 0.f; wb = cb; }
                        else if (role == 1) { wa = -ca + ta; wb = -tb; }
                        else { wa = -ta; wb = -cb + tb; }
                        fx -= de * (wa * A.x + wb * B.x);
                        fy -= de * (wa * A.y + wb * B.y);
                        fz -= de * (wa * A.z + wb * B.z);
                        if (ENERGY && role == 0) e_dih += (double)r.prm[0] * (1.0 + (double)cn);
                    }
                }
            }
            fx = quad_xadd<0xB1>(fx); fy = quad_xadd<0xB1>(fy); fz = quad_xadd<0xB1>(fz);   // lane ^ 1
            fx = quad_xadd<0x4E>(fx); fy = quad_xadd<0x4E>(fy); fz = quad_xadd<0x4E>(fz);   // lane ^ 2
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
        double v[7] = {e_bond, e_angle, e_dih, e_lj14, e_c14, e_rec, e_vir};
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

__global__ void add_ext_kernel(uint32_t S, const uint32_t* __restrict__ orig_of, const float4* __restrict__ ext,
                               float4* __restrict__ force, const uint32_t* gate, uint32_t thr_bits) {
    if (gate && *gate > thr_bits) return;
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    uint32_t o = orig_of[s];
    if (o == MDX_INVALID) return;
    float4 f = force[s], e = ext[o];
    f.x += e.x; f.y += e.y; f.z += e.z;
    force[s] = f;
}

int mdx_launch_bonded(mdx_handle* h, bool energy, const uint32_t* d_gate, uint32_t thr_bits) {
    const bool skip_bonded = (h->cfg.overrides & MDX_OVR_BONDED_DISABLED) != 0;
    if (skip_bonded && !h->pme_on) return MDX_OK;
    if (!h->n_roles) return MDX_OK;
    BondedArgs a{};
    a.S = h->S; a.role_off = h->d.role_off_s; a.roles = h->d.role_rec_s;
    a.posq = h->d.posq; a.force = h->d.force; a.energy = h->d.energy; a.gate = d_gate; a.thr_bits = thr_bits;
    for (int d = 0; d < 3; ++d) {
        a.p.box[d] = h->per[d] ? (h->box_hi[d] - h->box_lo[d]) : 0.f;
        a.p.inv_box[d] = h->per[d] ? 1.0f / a.p.box[d] : 0.f;
    }
    a.p.ewald_beta = h->cfg.ewald_alpha; a.p.skip_bonded = skip_bonded ? 1 : 0;
    mdx_prof_begin(h, energy ? 3 : 1);
    const dim3 g((uint32_t)(((size_t)h->S * BONDED_LPA + 255) / 256)), b(256);
    if (energy) hipLaunchKernelGGL(bonded_gather_kernel<true>, g, b, 0, h->stream, a);
    else hipLaunchKernelGGL(bonded_gather_kernel<false>, g, b, 0, h->stream, a);
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_add_ext(mdx_handle* h, const uint32_t* d_gate, uint32_t thr_bits) {
    if (!h->have_ext) return MDX_OK;
    hipLaunchKernelGGL(add_ext_kernel, dim3((h->S + 255) / 256), dim3(256), 0, h->stream, h->S, h->d.orig_of,
                       h->d.ext_orig, h->d.force, d_gate, thr_bits);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}
