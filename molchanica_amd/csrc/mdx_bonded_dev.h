// mdx_bonded_dev.h - evaluation of one bonded "role" (one term seen from one of its atoms); shared by the
// bonded gather kernel (mdx_bonded.hip) and the fused tail of the pair kernel (mdx_nonbonded.hip).
#pragma once
#include "mdx_internal.h"

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


// One launch per step (round 6, mdx_nonbonded_impl.h "STEP"): between two chunk boundaries the position array holds Y = x + dt v (the drift
// without the last kick) and the force array's .w holds dt^2 * 418.4 / m, so that EVERY reader of an atom reconstructs its position as
// Y + w F from two rows - the owner, the pair kernel's j-side and the bonded partners alike, with this one expression (bitwise the same).
__device__ __forceinline__ float4 step_pos(const float4 y, const float4 f) {
    return make_float4(fmaf(f.w, f.x, y.x), fmaf(f.w, f.y, y.y), fmaf(f.w, f.z, y.z), y.w);
}

struct RoleEnergies { double bond = 0.0, angle = 0.0, dih = 0.0, lj14 = 0.0, c14 = 0.0, rec = 0.0, vir = 0.0; };

// Adds to (fx, fy, fz) the force of role `r` on its own atom (position `self`).  ENERGY: the term's energy and
// virial are credited once, by the atom in role 0.
// role_compute: the arithmetic, with the term's first two partners (q0 = posq[r.p[0]], q1 = posq[r.p[1]]) already loaded -
// the fused bonded + kick + drift pass issues the partner loads of an atom's first roles together, ahead of the arithmetic.
// NODIH: the caller knows the system has bond and angle roles only (a box of flexible water): the dihedral branch - a third of the
// function's registers -, the 1-4 pairs and the Ewald exclusion corrections are compiled out (the fused bonded + kick + drift pass:
// 93 -> 67 VGPRs, five -> seven waves per SIMD)
template <bool ENERGY, bool NODIH = false>
__device__ __forceinline__ void role_compute(const RoleRec& r, const float4 prm4, const float4 self, const float4 q0, const float4 q1,
                                             const float4* __restrict__ posq,
                                             const BondedParams& p, float& fx, float& fy, float& fz, RoleEnergies& en,
                                             const float4* __restrict__ fstep = nullptr);

template <bool ENERGY, bool NODIH = false>
__device__ __forceinline__ void role_eval(const RoleRec& r, const float4* __restrict__ prm_tab, const float4 self,
                                          const float4* __restrict__ posq,
                                          const BondedParams& p, float& fx, float& fy, float& fz, RoleEnergies& en) {
    const uint32_t kind = r.meta & 0xFu;
    const float4 q0 = posq[r.p[0]];
    const float4 q1 = (kind == ROLE_ANGLE || kind == ROLE_DIHEDRAL) ? posq[r.p[1]] : q0;
    role_compute<ENERGY, NODIH>(r, prm_tab[r.meta >> 8], self, q0, q1, posq, p, fx, fy, fz, en);
}

// (positions in the step form - posq = Y, fstep = the force rows beside it: role_eval_step)
template <bool ENERGY>
__device__ __forceinline__ void role_eval_step(const RoleRec& r, const float4* __restrict__ prm_tab, const float4 self,
                                               const float4* __restrict__ y, const float4* __restrict__ fstep,
                                               const BondedParams& p, float& fx, float& fy, float& fz, RoleEnergies& en) {
    const uint32_t kind = r.meta & 0xFu;
    const float4 q0 = step_pos(y[r.p[0]], fstep[r.p[0]]);
    const float4 q1 = (kind == ROLE_ANGLE || kind == ROLE_DIHEDRAL) ? step_pos(y[r.p[1]], fstep[r.p[1]]) : q0;
    role_compute<ENERGY>(r, prm_tab[r.meta >> 8], self, q0, q1, y, p, fx, fy, fz, en, fstep);
}

template <bool ENERGY, bool NODIH>
__device__ __forceinline__ void role_compute(const RoleRec& r, const float4 prm4, const float4 self, const float4 q0, const float4 q1,
                                             const float4* __restrict__ posq,
                                             const BondedParams& p, float& fx, float& fy, float& fz, RoleEnergies& en,
                                             const float4* __restrict__ fstep) {
    const float prm[3] = {prm4.x, prm4.y, prm4.z};
    double& e_bond = en.bond; double& e_angle = en.angle; double& e_dih = en.dih; double& e_lj14 = en.lj14;
    double& e_c14 = en.c14; double& e_rec = en.rec; double& e_vir = en.vir;
    const uint32_t kind = r.meta & 0xFu, role = (r.meta >> 4) & 0xFu;
    if (!NODIH && kind == ROLE_EWALD_EXCL) {
        // the reciprocal sum includes this excluded / 1-4 pair: take erf(beta r)/r out again
        const float3 d = mimg(sub3(self, q0), p);
        const float r2 = dot3(d, d), rinv = rsqrtf(r2), rr = r2 * rinv, br = p.ewald_beta * rr;
        const float er = erff(br);
        const float fs = -prm[0] * (er * rinv - 1.1283791671f * p.ewald_beta * __expf(-br * br)) * rinv * rinv;
        fx += fs * d.x; fy += fs * d.y; fz += fs * d.z;
        if (ENERGY && role == 0) { e_rec -= (double)(prm[0] * er * rinv); e_vir += (double)(fs * r2); }
        return;
    }
    if (p.skip_bonded) return;
    if (kind == ROLE_BOND || kind == ROLE_PAIR14) {
        const float3 d = mimg(sub3(self, q0), p);
        const float r2 = dot3(d, d);
        float fs;
        if (NODIH || kind == ROLE_BOND) {
            const float rr = sqrtf(r2), dr = rr - prm[1];
            fs = -2.0f * prm[0] * dr / rr;
            if (ENERGY && role == 0) e_bond += (double)prm[0] * dr * dr;
        } else {   // prm: sigma_ij, 4*scale*eps_ij, scale*ke*qi*qj
            const float rinv = rsqrtf(r2), rinv2 = rinv * rinv;
            const float s2 = prm[0] * prm[0] * rinv2, s6 = s2 * s2 * s2;
            fs = (6.0f * prm[1] * s6 * (2.0f * s6 - 1.0f) + prm[2] * rinv) * rinv2;
            if (ENERGY && role == 0) {
                e_lj14 += (double)(prm[1] * s6 * (s6 - 1.0f));
                e_c14 += (double)(prm[2] * rinv);
            }
        }
        fx += fs * d.x; fy += fs * d.y; fz += fs * d.z;
        // virial: central pair forces only - angle and dihedral energies do not change
        // under a uniform scaling, their sum r_i . F_i is identically zero
        if (ENERGY && role == 0) e_vir += (double)(fs * r2);
    } else if (kind == ROLE_ANGLE) {
        // ordered atoms i - j(apex) - k; this lane is atom `role`
        const float4 pi = role == 0 ? self : q0;
        const float4 pj = role == 1 ? self : (role == 0 ? q0 : q1);
        const float4 pk = role == 2 ? self : q1;
        const float3 v1 = mimg(sub3(pi, pj), p), v2 = mimg(sub3(pk, pj), p);
        const float ir1 = rsqrtf(dot3(v1, v1)), ir2 = rsqrtf(dot3(v2, v2));
        float cs = dot3(v1, v2) * ir1 * ir2;
        cs = fminf(1.0f, fmaxf(-1.0f, cs));
        // theta from atan2(|v1 x v2|, v1 . v2): in fp32 acos(cs) loses half its digits near a straight angle
        // (d theta / d cs = -1 / sin theta), and sqrt(1 - cs^2) all of them; GAFF has theta0 = 180 deg (nitriles, alkynes)
        // and the synthetic helix of config 4 has 177 deg - with acos its angle forces carried ~2 kcal/mol/A of rounding
        // noise per step and the box heated by 5 % of its kinetic energy per 1000 steps
        const float3 cr = cross3(v1, v2);
        const float cl = sqrtf(dot3(cr, cr));
        const float th = atan2f(cl, dot3(v1, v2)), dth = th - prm[1];
        const float sn = fmaxf(cl * ir1 * ir2, 1e-6f);
        const float de = 2.0f * prm[0] * dth;   // dE/dtheta
        // dtheta/dr_i = -(v2/|v2| - cos v1/|v1|) / (|v1| sin)
        const float ci = de * ir1 / sn, ck = de * ir2 / sn;
        const float3 u1 = scale3(v1, ir1), u2 = scale3(v2, ir2);
        const float3 fi = make_float3(ci * (u2.x - cs * u1.x), ci * (u2.y - cs * u1.y), ci * (u2.z - cs * u1.z));
        const float3 fk = make_float3(ck * (u1.x - cs * u2.x), ck * (u1.y - cs * u2.y), ck * (u1.z - cs * u2.z));
        if (role == 0) { fx += fi.x; fy += fi.y; fz += fi.z; }
        else if (role == 2) { fx += fk.x; fy += fk.y; fz += fk.z; }
        else { fx -= fi.x + fk.x; fy -= fi.y + fk.y; fz -= fi.z + fk.z; }
        if (ENERGY && role == 0) e_angle += (double)prm[0] * dth * dth;
    } else if (!NODIH) {   // ROLE_DIHEDRAL: ordered atoms 0-1-2-3, this lane is atom `role`
        float4 q2 = posq[r.p[2]];
        if (fstep) q2 = step_pos(q2, fstep[r.p[2]]);
        const float4 p0 = role == 0 ? self : q0;
        const float4 p1 = role == 1 ? self : (role == 0 ? q0 : q1);
        const float4 p2 = role == 2 ? self : (role < 2 ? q1 : q2);
        const float4 p3 = role == 3 ? self : q2;
        // Blondel-Karplus: F = r0-r1, G = r1-r2, H = r3-r2, A = F x G, B = H x G
        const float3 F = mimg(sub3(p0, p1), p), G = mimg(sub3(p1, p2), p), H = mimg(sub3(p3, p2), p);
        const float3 A = cross3(F, G), B = cross3(H, G);
        const float A2 = dot3(A, A), B2 = dot3(B, B), G2 = dot3(G, G);
        if (A2 > 1e-12f && B2 > 1e-12f && G2 > 1e-12f) {
            const float Gn = sqrtf(G2), iGn = 1.0f / Gn;
            const float cosphi = dot3(A, B);
            const float sinphi = dot3(cross3(B, A), G) * iGn;
            const float phi = atan2f(sinphi, cosphi);
            float sn, cn;
            sincosf(prm[2] * phi - prm[1], &sn, &cn);
            const float de = -prm[0] * prm[2] * sn;   // dE/dphi
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
            if (ENERGY && role == 0) e_dih += (double)prm[0] * (1.0 + (double)cn);
        }
    }
}
