// mdx_bonded.hip — Amber bonded terms (bond / angle / dihedral) and scaled 1-4 pairs.
//
// Energy forms per the reference's parameter records, /root/reference src/ui/popup/ff_params.rs:
//   bond     E = k_b (r - r_0)^2                         :352-372
//   angle    E = k (theta - theta_0)^2                   :401-421
//   dihedral E = (barrier_height/divider) [1 + cos(periodicity*phi - phase)]   :474-511
// One term per lane over ONE flat index space (bonds | angles | dihedrals | 1-4 pairs) so the
// whole bonded phase is a single launch.  Term indices are kept in slot space (re-mapped at every
// neighbour rebuild), so the atoms of a term sit in the same or an adjacent tile: the position
// gathers and the f32 atomic force adds hit the same few cache lines.  HBM-bound (streaming index
// + parameter arrays, ~36 + 16 t B per atom-step).
#include "mdx_internal.h"

struct BondedArgs {
    uint32_t nb, na, nd, np;
    const uint32_t* bond; const float2* bond_p;
    const uint32_t* angle; const float2* angle_p;
    const uint32_t* dih; const float4* dih_p;
    const uint32_t* p14; const float4* p14_p;
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
__device__ __forceinline__ void add_force(float4* f, uint32_t s, float3 v) {
    atomicAdd(&f[s].x, v.x); atomicAdd(&f[s].y, v.y); atomicAdd(&f[s].z, v.z);
}
__device__ __forceinline__ float3 scale3(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }

template <bool ENERGY>
__global__ __launch_bounds__(256) void bonded_kernel(BondedArgs a) {
    if (a.gate && *a.gate > a.thr_bits) return;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0, e2 = 0.0;  // e2: Coulomb part of a 1-4 pair
    int kind = -1;
    if (i < a.nb) {
        kind = EN_BOND;
        const uint32_t s0 = a.bond[2 * i], s1 = a.bond[2 * i + 1];
        const float2 prm = a.bond_p[i];
        const float3 d = mimg(sub3(a.posq[s0], a.posq[s1]), a.p);
        const float r = sqrtf(dot3(d, d));
        const float dr = r - prm.y;
        const float fs = -2.0f * prm.x * dr / r;
        add_force(a.force, s0, scale3(d, fs));
        add_force(a.force, s1, scale3(d, -fs));
        if (ENERGY) e = (double)prm.x * dr * dr;
    } else if ((i -= a.nb) < a.na) {
        kind = EN_ANGLE;
        const uint32_t s0 = a.angle[3 * i], s1 = a.angle[3 * i + 1], s2 = a.angle[3 * i + 2];
        const float2 prm = a.angle_p[i];
        const float4 pj = a.posq[s1];
        const float3 v1 = mimg(sub3(a.posq[s0], pj), a.p), v2 = mimg(sub3(a.posq[s2], pj), a.p);
        const float ir1 = rsqrtf(dot3(v1, v1)), ir2 = rsqrtf(dot3(v2, v2));
        float cs = dot3(v1, v2) * ir1 * ir2;
        cs = fminf(1.0f, fmaxf(-1.0f, cs));
        const float th = acosf(cs), dth = th - prm.y;
        const float sn = fmaxf(sqrtf(1.0f - cs * cs), 1e-6f);
        const float de = 2.0f * prm.x * dth;          // dE/dtheta
        // dtheta/dr_i = -(v2/|v2| - cos v1/|v1|) / (|v1| sin)
        const float ci = de * ir1 / sn, ck = de * ir2 / sn;
        const float3 u1 = scale3(v1, ir1), u2 = scale3(v2, ir2);
        const float3 fi = make_float3(ci * (u2.x - cs * u1.x), ci * (u2.y - cs * u1.y), ci * (u2.z - cs * u1.z));
        const float3 fk = make_float3(ck * (u1.x - cs * u2.x), ck * (u1.y - cs * u2.y), ck * (u1.z - cs * u2.z));
        add_force(a.force, s0, fi);
        add_force(a.force, s2, fk);
        add_force(a.force, s1, make_float3(-fi.x - fk.x, -fi.y - fk.y, -fi.z - fk.z));
        if (ENERGY) e = (double)prm.x * dth * dth;
    } else if ((i -= a.na) < a.nd) {
        kind = EN_DIHEDRAL;
        const uint32_t s0 = a.dih[4 * i], s1 = a.dih[4 * i + 1], s2 = a.dih[4 * i + 2], s3 = a.dih[4 * i + 3];
        const float4 prm = a.dih_p[i];  // v, phase, n
        const float4 p0 = a.posq[s0], p1 = a.posq[s1], p2 = a.posq[s2], p3 = a.posq[s3];
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
            sincosf(prm.z * phi - prm.y, &sn, &cn);
            const float de = -prm.x * prm.z * sn;     // dE/dphi
            const float iA2 = 1.0f / A2, iB2 = 1.0f / B2;
            const float FG = dot3(F, G), HG = dot3(H, G);
            const float ca = -Gn * iA2, cb = Gn * iB2;
            const float ta = FG * iA2 * iGn, tb = HG * iB2 * iGn;
            // f_x = -dE/dphi * dphi/dr_x
            const float3 f0 = scale3(A, -de * ca);
            const float3 f3 = scale3(B, -de * cb);
            const float3 f1 = make_float3(-de * ((-ca + ta) * A.x - tb * B.x), -de * ((-ca + ta) * A.y - tb * B.y),
                                          -de * ((-ca + ta) * A.z - tb * B.z));
            const float3 f2 = make_float3(-de * ((-cb + tb) * B.x - ta * A.x), -de * ((-cb + tb) * B.y - ta * A.y),
                                          -de * ((-cb + tb) * B.z - ta * A.z));
            add_force(a.force, s0, f0); add_force(a.force, s1, f1);
            add_force(a.force, s2, f2); add_force(a.force, s3, f3);
            if (ENERGY) e = (double)prm.x * (1.0 + (double)cn);
        }
    } else if ((i -= a.nd) < a.np) {
        kind = EN_LJ14;
        const uint32_t s0 = a.p14[2 * i], s1 = a.p14[2 * i + 1];
        const float4 prm = a.p14_p[i];  // sigma_ij, 4*scale*eps_ij, scale*ke*qi*qj
        const float3 d = mimg(sub3(a.posq[s0], a.posq[s1]), a.p);
        const float r2 = dot3(d, d), rinv = rsqrtf(r2), rinv2 = rinv * rinv;
        const float s2 = prm.x * prm.x * rinv2, s6 = s2 * s2 * s2;
        const float fs = (6.0f * prm.y * s6 * (2.0f * s6 - 1.0f) + prm.z * rinv) * rinv2;
        add_force(a.force, s0, scale3(d, fs));
        add_force(a.force, s1, scale3(d, -fs));
        if (ENERGY) {
            e = (double)(prm.y * s6 * (s6 - 1.0f));
            e2 = (double)(prm.z * rinv);
        }
    }
    if (ENERGY) {
        // a wave may straddle two term kinds: reduce each kind separately
#pragma unroll
        for (int k = EN_BOND; k <= EN_DIHEDRAL; ++k) {
            double t = (kind == k) ? e : 0.0;
            if (__any(kind == k)) {
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m);
                if ((threadIdx.x & 63) == 0) atomicAdd(&a.energy[k], t);
            }
        }
        if (__any(kind == EN_LJ14)) {
            double t = (kind == EN_LJ14) ? e : 0.0, u = (kind == EN_LJ14) ? e2 : 0.0;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) { t += __shfl_xor(t, m); u += __shfl_xor(u, m); }
            if ((threadIdx.x & 63) == 0) { atomicAdd(&a.energy[EN_LJ14], t); atomicAdd(&a.energy[EN_COUL14], u); }
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
    if (h->cfg.overrides & MDX_OVR_BONDED_DISABLED) return MDX_OK;
    BondedArgs a{};
    a.nb = h->n_bonds; a.na = h->n_angles; a.nd = h->n_dih; a.np = h->n_p14;
    const uint32_t total = a.nb + a.na + a.nd + a.np;
    if (!total) return MDX_OK;
    a.bond = h->d.bond_s; a.bond_p = h->d.bond_p; a.angle = h->d.angle_s; a.angle_p = h->d.angle_p;
    a.dih = h->d.dih_s; a.dih_p = h->d.dih_p; a.p14 = h->d.p14_s; a.p14_p = h->d.p14_p;
    a.posq = h->d.posq; a.force = h->d.force; a.energy = h->d.energy; a.gate = d_gate; a.thr_bits = thr_bits;
    for (int d = 0; d < 3; ++d) {
        a.p.box[d] = h->periodic ? (h->box_hi[d] - h->box_lo[d]) : 0.f;
        a.p.inv_box[d] = h->periodic ? 1.0f / a.p.box[d] : 0.f;
    }
    mdx_prof_begin(h, 1);
    const dim3 g((total + 255) / 256), b(256);
    if (energy) hipLaunchKernelGGL(bonded_kernel<true>, g, b, 0, h->stream, a);
    else hipLaunchKernelGGL(bonded_kernel<false>, g, b, 0, h->stream, a);
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
