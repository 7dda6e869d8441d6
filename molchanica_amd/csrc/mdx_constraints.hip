// mdx_constraints.hip — holonomic distance constraints (SHAKE / RATTLE) and massless virtual
// sites.
//
// The reference's default operating point is dt = 2 fs with constrained hydrogens and 4-site OPC
// water: `HydrogenConstraint::{Shake{shake_tolerance}, Linear{order,iter}, Flexible}`
// (/root/reference src/ui/panels/md.rs:362-371), `md.water[i].{o,h0,h1,m}`
// (src/properties/sol_shrinking_box.rs:605-613).  The solver itself lives in the absent crate; what
// is built here (and restated by the oracle):
//   * constraints are grouped into connected clusters of <= 4 atoms / <= 6 constraints — a rigid
//     water (O-H, O-H, H-H) or a heavy atom with up to three hydrogens — or a heavy atom with FOUR hydrogens (ConsStar5:
//     ammonium, methane), one lane per cluster, the whole
//     Gauss-Seidel SHAKE iteration in registers.  Clusters are independent, so there is no
//     inter-lane traffic and no atomics; a cluster's atoms sit in the same or an adjacent tile.
//   * position stage after every kick+drift: x(t) is rebuilt as x' - dt v', the SHAKE corrections
//     act along the OLD bond vectors, and v' += (x'' - x')/dt  (leap-frog form inside a burst);
//   * velocity stage (RATTLE) after the closing half kick of a burst, so velocities handed to the
//     caller have no component along a constrained bond;
//   * virtual site r = r0 + a (r1 - r0) + b (r2 - r0): constructed after every position update,
//     its force spread onto the three parents with the same weights (single writer per parent:
//     a parent belongs to one site), the site itself carries no mass and is never integrated.
// Streaming kernels, HBM/latency-bound; ~0.7 Gflop per step at 343 k waters, noise next to the pair loop.
#include "mdx_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

#define FAIL(code, msg) do { mdx_set_error(msg); return (code); } while (0)
static inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

struct ConsParams { float box[3], inv_box[3]; float tol; int max_iter; int settle; float vir_scale; };

__device__ __forceinline__ float3 mimg3(float3 d, const ConsParams& p) {
    if (p.box[0] > 0.f) d.x -= rintf(d.x * p.inv_box[0]) * p.box[0];
    if (p.box[1] > 0.f) d.y -= rintf(d.y * p.inv_box[1]) * p.box[1];
    if (p.box[2] > 0.f) d.z -= rintf(d.z * p.inv_box[2]) * p.box[2];
    return d;
}

// ---- rigid three-site water: analytic solution (SETTLE, Miyamoto & Kollman 1992) -----------------------------------
// A cluster of three atoms with all three distances fixed, two equal legs from atom 0 and equal masses on atoms 1, 2
// (every rigid water; OPC's M site is a virtual site, not a member) needs no iteration: the constrained triangle is the
// canonical one rotated by three angles that follow in closed form from the old positions (which satisfy the
// constraints) and the unconstrained new ones.  The Gauss-Seidel SHAKE sweep it replaces converges linearly (~12 sweeps to
// 1e-5 in fp32, each a chain of dependent operations): at 1 M sites of rigid OPC the position stage took 82 us and the
// velocity stage 68 us of a 1.05 ms step.  Other clusters (X-H groups of a solute) keep SHAKE / RATTLE.
struct Triangle { float l01, l12; bool ok; };
__device__ __forceinline__ Triangle rigid_triangle(const ConsGroup& cg, const float im[4]) {
    Triangle t{0.f, 0.f, false};
    if (cg.natoms != 3 || cg.ncons != 3 || !(im[0] > 0.f) || !(im[1] > 0.f) || im[1] != im[2]) return t;
    float l01 = 0.f, l02 = 0.f, l12 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int a = min((int)cg.ca[c], (int)cg.cb[c]), b = max((int)cg.ca[c], (int)cg.cb[c]);
        if (a == 0 && b == 1) l01 = cg.len[c];
        else if (a == 0 && b == 2) l02 = cg.len[c];
        else if (a == 1 && b == 2) l12 = cg.len[c];
    }
    t.l01 = l01; t.l12 = l12;
    t.ok = l01 > 0.f && l02 > 0.f && l12 > 0.f && fabsf(l01 - l02) <= 1e-6f * l01 && l12 < 2.0f * l01;
    return t;
}
__device__ __forceinline__ float3 cross3f(float3 a, float3 b) { return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float dot3f(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 unit3f(float3 a) { const float s = rsqrtf(dot3f(a, a)); return make_float3(a.x * s, a.y * s, a.z * s); }
// xo: old positions (constraints satisfied), xn: unconstrained new positions, both relative to one origin; on success xn
// holds the constrained positions.  false: degenerate input (a site moved by about a bond length) - the caller iterates.
__device__ __forceinline__ bool settle_positions(const float3 xo[4], float3 xn[4], const float im[4], const Triangle& t) {
    const float mO = 1.0f / im[0], mH = 1.0f / im[1], iM = 1.0f / (mO + 2.0f * mH);
    const float rc = 0.5f * t.l12, hgt = sqrtf(t.l01 * t.l01 - rc * rc), ra = hgt * 2.0f * mH * iM, rb = hgt - ra;
    const float3 b0 = make_float3(xo[1].x - xo[0].x, xo[1].y - xo[0].y, xo[1].z - xo[0].z);
    const float3 c0 = make_float3(xo[2].x - xo[0].x, xo[2].y - xo[0].y, xo[2].z - xo[0].z);
    const float wO = mO * iM, wH = mH * iM;
    const float3 com = make_float3(wO * xn[0].x + wH * (xn[1].x + xn[2].x), wO * xn[0].y + wH * (xn[1].y + xn[2].y),
                                   wO * xn[0].z + wH * (xn[1].z + xn[2].z));
    const float3 a1 = make_float3(xn[0].x - com.x, xn[0].y - com.y, xn[0].z - com.z);
    const float3 b1 = make_float3(xn[1].x - com.x, xn[1].y - com.y, xn[1].z - com.z);
    const float3 c1 = make_float3(xn[2].x - com.x, xn[2].y - com.y, xn[2].z - com.z);
    const float3 ez = unit3f(cross3f(b0, c0));
    const float3 ex = unit3f(cross3f(a1, ez));
    const float3 ey = cross3f(ez, ex);
    const float xb0d = dot3f(b0, ex), yb0d = dot3f(b0, ey), xc0d = dot3f(c0, ex), yc0d = dot3f(c0, ey);
    const float za1d = dot3f(a1, ez);
    const float xb1d = dot3f(b1, ex), yb1d = dot3f(b1, ey), zb1d = dot3f(b1, ez);
    const float xc1d = dot3f(c1, ex), yc1d = dot3f(c1, ey), zc1d = dot3f(c1, ez);
    const float sinphi = za1d / ra, cp2 = 1.0f - sinphi * sinphi;
    if (!(cp2 > 1e-6f)) return false;
    const float cosphi = sqrtf(cp2);
    const float sinpsi = (zb1d - zc1d) / (2.0f * rc * cosphi), cs2 = 1.0f - sinpsi * sinpsi;
    if (!(cs2 > 1e-6f)) return false;
    const float cospsi = sqrtf(cs2);
    const float ya2d = ra * cosphi, xb2d = -rc * cospsi, t1 = -rb * cosphi, t2 = rc * sinpsi * sinphi;
    const float yb2d = t1 - t2, yc2d = t1 + t2;
    const float alpha = xb2d * (xb0d - xc0d) + yb0d * yb2d + yc0d * yc2d;
    const float beta = xb2d * (yc0d - yb0d) + xb0d * yb2d + xc0d * yc2d;
    const float gamma = xb0d * yb1d - xb1d * yb0d + xc0d * yc1d - xc1d * yc0d;
    const float al2be2 = alpha * alpha + beta * beta, disc = al2be2 - gamma * gamma;
    if (!(disc > 0.f) || !(al2be2 > 0.f)) return false;
    const float sinthe = (alpha * gamma - beta * sqrtf(disc)) / al2be2, ct2 = 1.0f - sinthe * sinthe;
    if (!(ct2 > 1e-6f)) return false;
    const float costhe = sqrtf(ct2);
    const float xa3 = -ya2d * sinthe, ya3 = ya2d * costhe, za3 = za1d;
    const float xb3 = xb2d * costhe - yb2d * sinthe, yb3 = xb2d * sinthe + yb2d * costhe, zb3 = zb1d;
    const float xc3 = -xb2d * costhe - yc2d * sinthe, yc3 = -xb2d * sinthe + yc2d * costhe, zc3 = zc1d;
    xn[0] = make_float3(com.x + xa3 * ex.x + ya3 * ey.x + za3 * ez.x, com.y + xa3 * ex.y + ya3 * ey.y + za3 * ez.y, com.z + xa3 * ex.z + ya3 * ey.z + za3 * ez.z);
    xn[1] = make_float3(com.x + xb3 * ex.x + yb3 * ey.x + zb3 * ez.x, com.y + xb3 * ex.y + yb3 * ey.y + zb3 * ez.y, com.z + xb3 * ex.z + yb3 * ey.z + zb3 * ez.z);
    xn[2] = make_float3(com.x + xc3 * ex.x + yc3 * ey.x + zc3 * ez.x, com.y + xc3 * ex.y + yc3 * ey.y + zc3 * ez.y, com.z + xc3 * ex.z + yc3 * ey.z + zc3 * ez.z);
    return true;
}

// The constraints of a cluster as a table over the six atom pairs of up to four atoms, in the fixed order (0,1) (0,2) (0,3) (1,2)
// (1,3) (2,3): length of the constraint on that pair, 0 = none.  The sweeps below then walk the PAIRS with compile-time atom
// indices; walking the constraints (cg.ca[c], cg.cb[c]) indexes the local position arrays with run-time values, which puts them
// into scratch memory (64-112 B per lane, and every access of the sweep a memory instruction).
__device__ __forceinline__ void pair_lengths(const ConsGroup& cg, float L[6]) {
#pragma unroll
    for (int q = 0; q < 6; ++q) L[q] = 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        if (c < (int)cg.ncons) {
            const int a = min((int)cg.ca[c], (int)cg.cb[c]), b = max((int)cg.ca[c], (int)cg.cb[c]);
            const int q = a == 0 ? b - 1 : (a == 1 ? b + 1 : 5);      // (0,1) 0  (0,2) 1  (0,3) 2  (1,2) 3  (1,3) 4  (2,3) 5
#pragma unroll
            for (int k = 0; k < 6; ++k) L[k] = (k == q) ? cg.len[c] : L[k];
        }
    }
}
#define MDX_PAIR_A(q) ((q) < 3 ? 0 : ((q) < 5 ? 1 : 2))
#define MDX_PAIR_B(q) ((q) < 3 ? (q) + 1 : ((q) < 5 ? (q) - 1 : 3))

// ---- SHAKE: positions ---------------------------------------------------------------------------
// RIGID3: every cluster of the handle is a rigid three-site water (host-checked): constraint pairs are (0,1), (0,2), (1,2) by
// construction, so nothing indexes the local position arrays with a run-time value and they live in registers - the general
// flavour keeps them in scratch memory (112 B per lane) because its SHAKE sweep addresses them through cg.ca / cg.cb.
template <bool RIGID3>
__global__ __launch_bounds__(128) void constrain_positions_kernel(uint32_t n_groups, const ConsGroup* __restrict__ groups,
                                                                  float4* __restrict__ posq, float4* __restrict__ vel,
                                                                  float4* __restrict__ ref, float dt, ConsParams p,
                                                                  float* __restrict__ cons_vir,
                                                                  const uint32_t* gate, uint32_t* disp_out, uint32_t thr,
                                                                  uint32_t* prune_out, float path_thr, const uint32_t* __restrict__ n_dev,
                                                                  const GroupSite* __restrict__ gsite, uint32_t skip_wstep) {
    if (gate && *gate > thr) return;
    if (n_dev) n_groups = min(n_groups, *n_dev);      // (clusters in slot order: the ones this handle solves come first)
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float d2max = 0.f, pmax = 0.f;   // pmax: dual pair list, the longest path since the last pruning pass (ref[].w)
    // (skip_wstep: this step's water_step_kernel has taken the rigid waters through kick, drift and SETTLE - the rest is solved here)
    if (g < n_groups && !(skip_wstep && groups[g].wstep)) {
        const ConsGroup cg = groups[g];
        // local frame: displacement of every atom from atom 0 (minimum image), new and old
        float3 xn[4], xo[4];
        float im[4];
        float4 p0 = make_float4(0, 0, 0, 0);
        float4 pk_in[4], vk_in[4];      // kept for the write-back (it re-read both rows: 6 of the 22 scattered loads per water)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xn[k] = make_float3(0, 0, 0); xo[k] = make_float3(0, 0, 0); im[k] = 0.f;
            pk_in[k] = make_float4(0, 0, 0, 0); vk_in[k] = pk_in[k];
            if (k < (int)cg.natoms) {
                const float4 pk = posq[cg.atom[k]], vk = vel[cg.atom[k]];
                pk_in[k] = pk; vk_in[k] = vk;
                if (k == 0) p0 = pk;
                xn[k] = mimg3(make_float3(pk.x - p0.x, pk.y - p0.y, pk.z - p0.z), p);
                xo[k] = make_float3(xn[k].x - dt * vk.x, xn[k].y - dt * vk.y, xn[k].z - dt * vk.z);
                im[k] = vk.w;   // 418.4 / m: only ratios matter
            }
        }
        const float3 x0 = xn[0];
        float3 xs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[k] = xn[k];
        // Constraint virial.  A correction gk*im[a]*r of atom a is the work of a force G = 2 gk r / dt^2
        // (kcal/mol/A: im = 418.4/m, and the force acts through the half kick dt/2 and the drift dt) along
        // the OLD bond vector r; its contribution to sum r_i . F_i is G . (r_a - r_b) = 2 gk |r|^2 / dt^2.
        float wc = 0.f;
        bool settled = false;
        const Triangle tri = rigid_triangle(cg, im);
        if (p.settle && dt != 0.f) {      // (dt = 0 is the projection of caller-supplied geometry: its "old" positions are not on the constraints)
            if (tri.ok && settle_positions(xo, xn, im, tri)) {
                settled = true;
                // the virial sum of the equivalent pair corrections: sum_k g_k |r_k|^2 = sum_i (dx_i / im_i) . x_old_i
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    wc += ((xn[k].x - xs[k].x) * xo[k].x + (xn[k].y - xs[k].y) * xo[k].y + (xn[k].z - xs[k].z) * xo[k].z) / im[k];
            }
        }
        float Lq[6];
        if (RIGID3) { Lq[0] = tri.l01; Lq[1] = tri.l01; Lq[2] = 0.f; Lq[3] = tri.l12; Lq[4] = 0.f; Lq[5] = 0.f; }
        else pair_lengths(cg, Lq);
        for (int it = 0; it < ((settled || (RIGID3 && cg.natoms != 3)) ? 0 : p.max_iter); ++it) {
            bool done = true;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (RIGID3 && (q == 2 || q > 3)) continue;
                if (!(Lq[q] > 0.f)) continue;
                const int a = MDX_PAIR_A(q), b = MDX_PAIR_B(q);
                const float3 s = make_float3(xn[a].x - xn[b].x, xn[a].y - xn[b].y, xn[a].z - xn[b].z);
                const float l2 = Lq[q] * Lq[q];
                const float diff = l2 - (s.x * s.x + s.y * s.y + s.z * s.z);
                if (fabsf(diff) > 2.0f * p.tol * l2) {
                    done = false;
                    const float3 r = make_float3(xo[a].x - xo[b].x, xo[a].y - xo[b].y, xo[a].z - xo[b].z);
                    const float sr = s.x * r.x + s.y * r.y + s.z * r.z;
                    const float gk = diff / (2.0f * sr * (im[a] + im[b]));
                    wc += gk * (r.x * r.x + r.y * r.y + r.z * r.z);
                    xn[a].x += gk * im[a] * r.x; xn[a].y += gk * im[a] * r.y; xn[a].z += gk * im[a] * r.z;
                    xn[b].x -= gk * im[b] * r.x; xn[b].y -= gk * im[b] * r.y; xn[b].z -= gk * im[b] * r.z;
                }
            }
            if (done) break;
        }
        const float idt = dt != 0.f ? 1.0f / dt : 0.f;
        // (vir_scale: 2 when the correction answers the opening half kick alone - a force acting through dt/2 and the
        // drift - and 1 when kick and drift of a fused step carried the velocity projection with them: through dt)
        if (cons_vir) cons_vir[g] = p.vir_scale * wc * idt * idt;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= (int)cg.natoms) break;
            const float3 dx = make_float3(xn[k].x - xs[k].x, xn[k].y - xs[k].y, xn[k].z - xs[k].z);
            if (dx.x != 0.f || dx.y != 0.f || dx.z != 0.f) {
                float4 pk = pk_in[k], vk = vk_in[k];
                pk.x += dx.x; pk.y += dx.y; pk.z += dx.z;
                vk.x += dx.x * idt; vk.y += dx.y * idt; vk.z += dx.z * idt;
                posq[cg.atom[k]] = pk; vel[cg.atom[k]] = vk;
                const float4 r = ref[cg.atom[k]];
                const float ex = pk.x - r.x, ey = pk.y - r.y, ez = pk.z - r.z;
                d2max = fmaxf(d2max, ex * ex + ey * ey + ez * ez);
                if (prune_out) {   // the SHAKE correction lengthens the atom's path like the drift did
                    const float w = r.w + sqrtf(dx.x * dx.x + dx.y * dx.y + dx.z * dx.z);
                    ref[cg.atom[k]].w = w;
                    pmax = fmaxf(pmax, w);
                }
            }
        }
        (void)x0;
        if (gsite) {
            // the cluster's virtual site (GroupSite): r_p0 + a (r_p1 - r_p0) + b (r_p2 - r_p0) from the constrained positions in
            // registers, moved to the periodic image nearest to where the site is stored (see vsite_construct_kernel)
            const GroupSite gv = gsite[g];
            if (gv.on && gv.site != MDX_INVALID && cg.natoms >= 3) {
                // weights of the cluster's atoms in the site (per atom, by comparison: an indexed read of xn would put the array
                // into scratch memory)
                float3 fresh = make_float3(p0.x, p0.y, p0.z);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float wk = (k == (int)gv.k0 ? 1.0f - gv.a - gv.b : 0.f) + (k == (int)gv.k1 ? gv.a : 0.f) + (k == (int)gv.k2 ? gv.b : 0.f);
                    fresh.x += wk * xn[k].x; fresh.y += wk * xn[k].y; fresh.z += wk * xn[k].z;
                }
                float4 m = posq[gv.site];
                const float3 back = mimg3(make_float3(m.x - fresh.x, m.y - fresh.y, m.z - fresh.z), p);
                m.x -= back.x; m.y -= back.y; m.z -= back.z;
                posq[gv.site] = m;
            }
        }
    }
    if (disp_out) {
        if (!(d2max < 1.0e30f)) d2max = 3.0e38f;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) d2max = fmaxf(d2max, __shfl_xor(d2max, m));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(d2max) > thr) atomicMax(disp_out, __float_as_uint(d2max));
    }
    if (prune_out) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, m));
        if ((threadIdx.x & 63) == 0 && !(pmax <= path_thr)) *prune_out = 1u;
    }
}

// ---- RATTLE: velocities -----------------------------------------------------------------------------
template <bool RIGID3>      // (see constrain_positions_kernel)
__global__ __launch_bounds__(128) void constrain_velocities_kernel(uint32_t n_groups, const ConsGroup* __restrict__ groups,
                                                                   const float4* __restrict__ posq, float4* __restrict__ vel,
                                                                   ConsParams p, const uint32_t* gate, uint32_t thr,
                                                                   const uint32_t* __restrict__ n_dev) {
    if (gate && *gate > thr) return;
    if (n_dev) n_groups = min(n_groups, *n_dev);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const ConsGroup cg = groups[g];
    float3 x[4], v[4], v0[4];
    float im[4];
    float4 p0 = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = make_float3(0, 0, 0); v[k] = make_float3(0, 0, 0); im[k] = 0.f;
        if (k < (int)cg.natoms) {
            const float4 pk = posq[cg.atom[k]], vk = vel[cg.atom[k]];
            if (k == 0) p0 = pk;
            x[k] = mimg3(make_float3(pk.x - p0.x, pk.y - p0.y, pk.z - p0.z), p);
            v[k] = make_float3(vk.x, vk.y, vk.z); im[k] = vk.w;
        }
        v0[k] = v[k];
    }
    bool solved = false;
    if (p.settle && rigid_triangle(cg, im).ok) {
        // three multipliers from one 3 x 3 system: the corrected velocities v_i - im_i sum_k s_ik tau_k r_k must have no
        // component along any of the three bonds
        float3 r[3]; float rhs[3]; int ia[3], ib[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ia[c] = c == 2 ? 1 : 0; ib[c] = c == 0 ? 1 : 2;      // (a rigid triangle: its constraints are (0,1), (0,2), (1,2), whatever their order in the record)
            r[c] = make_float3(x[ia[c]].x - x[ib[c]].x, x[ia[c]].y - x[ib[c]].y, x[ia[c]].z - x[ib[c]].z);
            rhs[c] = dot3f(r[c], make_float3(v[ia[c]].x - v[ib[c]].x, v[ia[c]].y - v[ib[c]].y, v[ia[c]].z - v[ib[c]].z));
        }
        float A[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float sa = (ia[k] == ia[l] ? 1.f : 0.f) - (ia[k] == ib[l] ? 1.f : 0.f);    // sign of constraint l on atom ia[k]
                const float sb = (ib[k] == ia[l] ? 1.f : 0.f) - (ib[k] == ib[l] ? 1.f : 0.f);
                A[k][l] = dot3f(r[k], r[l]) * (im[ia[k]] * sa - im[ib[k]] * sb);
            }
        const float det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                          A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        if (fabsf(det) > 0.f) {
            const float id = 1.0f / det;
            float tau[3];
            tau[0] = id * (rhs[0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (rhs[1] * A[2][2] - A[1][2] * rhs[2]) +
                           A[0][2] * (rhs[1] * A[2][1] - A[1][1] * rhs[2]));
            tau[1] = id * (A[0][0] * (rhs[1] * A[2][2] - A[1][2] * rhs[2]) - rhs[0] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                           A[0][2] * (A[1][0] * rhs[2] - rhs[1] * A[2][0]));
            tau[2] = id * (A[0][0] * (A[1][1] * rhs[2] - rhs[1] * A[2][1]) - A[0][1] * (A[1][0] * rhs[2] - rhs[1] * A[2][0]) +
                           rhs[0] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int a = ia[c], b = ib[c];
                v[a].x -= tau[c] * im[a] * r[c].x; v[a].y -= tau[c] * im[a] * r[c].y; v[a].z -= tau[c] * im[a] * r[c].z;
                v[b].x += tau[c] * im[b] * r[c].x; v[b].y += tau[c] * im[b] * r[c].y; v[b].z += tau[c] * im[b] * r[c].z;
            }
            solved = true;
        }
    }
    float Lq[6];
    if (RIGID3) { Lq[0] = 1.f; Lq[1] = 1.f; Lq[2] = 0.f; Lq[3] = 1.f; Lq[4] = 0.f; Lq[5] = 0.f; }      // (which pairs; the lengths are read off the positions)
    else pair_lengths(cg, Lq);
    for (int it = 0; it < ((solved || (RIGID3 && cg.natoms != 3)) ? 0 : p.max_iter); ++it) {
        bool done = true;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (RIGID3 && (q == 2 || q > 3)) continue;
            if (!(Lq[q] > 0.f)) continue;
            const int a = MDX_PAIR_A(q), b = MDX_PAIR_B(q);
            const float3 s = make_float3(x[a].x - x[b].x, x[a].y - x[b].y, x[a].z - x[b].z);
            const float3 w = make_float3(v[a].x - v[b].x, v[a].y - v[b].y, v[a].z - v[b].z);
            const float dot = s.x * w.x + s.y * w.y + s.z * w.z;
            const float l2 = RIGID3 ? s.x * s.x + s.y * s.y + s.z * s.z : Lq[q] * Lq[q];      // (positions are on the constraints: |s| = the length)
            // |d/dt of the bond length| relative to 1 Å/ps-scale speeds
            if (fabsf(dot) > p.tol * l2 * 10.0f) {
                done = false;
                const float gk = dot / (l2 * (im[a] + im[b]));
                v[a].x -= gk * im[a] * s.x; v[a].y -= gk * im[a] * s.y; v[a].z -= gk * im[a] * s.z;
                v[b].x += gk * im[b] * s.x; v[b].y += gk * im[b] * s.y; v[b].z += gk * im[b] * s.z;
            }
        }
        if (done) break;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= (int)cg.natoms) break;
        if (v[k].x != v0[k].x || v[k].y != v0[k].y || v[k].z != v0[k].z) {
            float4 vk = vel[cg.atom[k]];
            vk.x = v[k].x; vk.y = v[k].y; vk.z = v[k].z;
            vel[cg.atom[k]] = vk;
        }
    }
}

// ---- X-H4 stars (ConsStar5): the same SHAKE / RATTLE sweeps over the four bonds of a heavy atom with four hydrogens ---------
// One lane per star, everything in registers (the bonds are (0, k): compile-time indices).  A record whose atom[0] is
// MDX_INVALID is not solved here (decomposed handle: the owner of the centre solves the star).
__global__ __launch_bounds__(128) void constrain_positions_star5_kernel(uint32_t n, const ConsStar5* __restrict__ stars, float4* __restrict__ posq,
                                                                        float4* __restrict__ vel, float4* __restrict__ ref, float dt, ConsParams p,
                                                                        float* __restrict__ cons_vir, const uint32_t* gate, uint32_t* disp_out,
                                                                        uint32_t thr, uint32_t* prune_out, float path_thr) {
    if (gate && *gate > thr) return;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float d2max = 0.f, pmax = 0.f;
    if (g < n) {
        const ConsStar5 cs = stars[g];
        if (cs.atom[0] != MDX_INVALID) {
            float3 xn[5], xo[5], xs[5];
            float im[5];
            float4 p0 = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float4 pk = posq[cs.atom[k]], vk = vel[cs.atom[k]];
                if (k == 0) p0 = pk;
                xn[k] = mimg3(make_float3(pk.x - p0.x, pk.y - p0.y, pk.z - p0.z), p);
                xo[k] = make_float3(xn[k].x - dt * vk.x, xn[k].y - dt * vk.y, xn[k].z - dt * vk.z);
                im[k] = vk.w; xs[k] = xn[k];
            }
            float wc = 0.f;
            for (int it = 0; it < p.max_iter; ++it) {
                bool done = true;
#pragma unroll
                for (int b = 1; b < 5; ++b) {
                    const float3 s = make_float3(xn[0].x - xn[b].x, xn[0].y - xn[b].y, xn[0].z - xn[b].z);
                    const float l2 = cs.len[b - 1] * cs.len[b - 1];
                    const float diff = l2 - (s.x * s.x + s.y * s.y + s.z * s.z);
                    if (fabsf(diff) > 2.0f * p.tol * l2) {
                        done = false;
                        const float3 r = make_float3(xo[0].x - xo[b].x, xo[0].y - xo[b].y, xo[0].z - xo[b].z);
                        const float sr = s.x * r.x + s.y * r.y + s.z * r.z;
                        const float gk = diff / (2.0f * sr * (im[0] + im[b]));
                        wc += gk * (r.x * r.x + r.y * r.y + r.z * r.z);
                        xn[0].x += gk * im[0] * r.x; xn[0].y += gk * im[0] * r.y; xn[0].z += gk * im[0] * r.z;
                        xn[b].x -= gk * im[b] * r.x; xn[b].y -= gk * im[b] * r.y; xn[b].z -= gk * im[b] * r.z;
                    }
                }
                if (done) break;
            }
            const float idt = dt != 0.f ? 1.0f / dt : 0.f;
            if (cons_vir) cons_vir[g] = p.vir_scale * wc * idt * idt;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float3 dx = make_float3(xn[k].x - xs[k].x, xn[k].y - xs[k].y, xn[k].z - xs[k].z);
                if (dx.x != 0.f || dx.y != 0.f || dx.z != 0.f) {
                    float4 pk = posq[cs.atom[k]], vk = vel[cs.atom[k]];
                    pk.x += dx.x; pk.y += dx.y; pk.z += dx.z;
                    vk.x += dx.x * idt; vk.y += dx.y * idt; vk.z += dx.z * idt;
                    posq[cs.atom[k]] = pk; vel[cs.atom[k]] = vk;
                    const float4 r = ref[cs.atom[k]];
                    const float ex = pk.x - r.x, ey = pk.y - r.y, ez = pk.z - r.z;
                    d2max = fmaxf(d2max, ex * ex + ey * ey + ez * ez);
                    if (prune_out) {
                        const float w = r.w + sqrtf(dx.x * dx.x + dx.y * dx.y + dx.z * dx.z);
                        ref[cs.atom[k]].w = w;
                        pmax = fmaxf(pmax, w);
                    }
                }
            }
        } else if (cons_vir) cons_vir[g] = 0.f;
    }
    if (disp_out) {
        if (!(d2max < 1.0e30f)) d2max = 3.0e38f;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) d2max = fmaxf(d2max, __shfl_xor(d2max, m));
        if ((threadIdx.x & 63) == 0 && __float_as_uint(d2max) > thr) atomicMax(disp_out, __float_as_uint(d2max));
    }
    if (prune_out) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, m));
        if ((threadIdx.x & 63) == 0 && !(pmax <= path_thr)) *prune_out = 1u;
    }
}

__global__ __launch_bounds__(128) void constrain_velocities_star5_kernel(uint32_t n, const ConsStar5* __restrict__ stars, const float4* __restrict__ posq,
                                                                         float4* __restrict__ vel, ConsParams p, const uint32_t* gate, uint32_t thr) {
    if (gate && *gate > thr) return;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const ConsStar5 cs = stars[g];
    if (cs.atom[0] == MDX_INVALID) return;
    float3 x[5], v[5], v0[5];
    float im[5];
    float4 p0 = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float4 pk = posq[cs.atom[k]], vk = vel[cs.atom[k]];
        if (k == 0) p0 = pk;
        x[k] = mimg3(make_float3(pk.x - p0.x, pk.y - p0.y, pk.z - p0.z), p);
        v[k] = make_float3(vk.x, vk.y, vk.z); v0[k] = v[k]; im[k] = vk.w;
    }
    for (int it = 0; it < p.max_iter; ++it) {
        bool done = true;
#pragma unroll
        for (int b = 1; b < 5; ++b) {
            const float3 s = make_float3(x[0].x - x[b].x, x[0].y - x[b].y, x[0].z - x[b].z);
            const float3 w = make_float3(v[0].x - v[b].x, v[0].y - v[b].y, v[0].z - v[b].z);
            const float dot = s.x * w.x + s.y * w.y + s.z * w.z;
            const float l2 = cs.len[b - 1] * cs.len[b - 1];
            if (fabsf(dot) > p.tol * l2 * 10.0f) {
                done = false;
                const float gk = dot / (l2 * (im[0] + im[b]));
                v[0].x -= gk * im[0] * s.x; v[0].y -= gk * im[0] * s.y; v[0].z -= gk * im[0] * s.z;
                v[b].x += gk * im[b] * s.x; v[b].y += gk * im[b] * s.y; v[b].z += gk * im[b] * s.z;
            }
        }
        if (done) break;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (v[k].x != v0[k].x || v[k].y != v0[k].y || v[k].z != v0[k].z) {
            float4 vk = vel[cs.atom[k]];
            vk.x = v[k].x; vk.y = v[k].y; vk.z = v[k].z;
            vel[cs.atom[k]] = vk;
        }
}

__global__ void remap_star5_kernel(uint32_t n, const ConsStar5* __restrict__ so, const uint32_t* __restrict__ slot_of, ConsStar5* __restrict__ ss,
                                   uint32_t* err, const uint8_t* __restrict__ slot_flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ConsStar5 g = so[i];
    if (slot_flags) {      // decomposed handle: the star is solved by the rank that owns its centre
        const uint32_t s0 = slot_of[g.atom[0]];
        if (s0 == MDX_INVALID || !(slot_flags[s0] & 2u)) { g.atom[0] = MDX_INVALID; ss[i] = g; return; }
    }
    for (int k = 0; k < 5; ++k) {
        const uint32_t s = slot_of[g.atom[k]];
        if (s == MDX_INVALID) atomicOr(err, 8u);
        g.atom[k] = s;
    }
    ss[i] = g;
}

// ---- virtual sites -----------------------------------------------------------------------------------
__global__ void vsite_construct_kernel(uint32_t n, const VSite* __restrict__ vs, float4* __restrict__ posq, ConsParams p,
                                       const uint32_t* gate, uint32_t thr) {
    if (gate && *gate > thr) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const VSite v = vs[i];
    if (v.site == MDX_INVALID) return;
    const float4 r0 = posq[v.p0], r1 = posq[v.p1], r2 = posq[v.p2];
    const float3 d1 = mimg3(make_float3(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z), p);
    const float3 d2 = mimg3(make_float3(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z), p);
    // The site goes to the periodic image of r0 + a d1 + b d2 that is nearest to where it is stored now: the pair
    // list holds ONE image shift per cluster pair, fixed at the rebuild, which wrapped every atom into the box on its
    // own - a site whose parent sits just across a box face must stay on ITS side of that face, where its cluster's
    // shifts expect it, as an integrated atom does by moving continuously.  (Placed next to the parent it sat a box
    // length away from the image the list had paired it with: every water straddling a face fed energy into the box,
    // 15 kcal/mol/ps per water at 23 k sites.  Minimum image per atom pair, as the oracle has, never sees this.)
    float4 m = posq[v.site];
    const float3 fresh = make_float3(r0.x + v.a * d1.x + v.b * d2.x, r0.y + v.a * d1.y + v.b * d2.y, r0.z + v.a * d1.z + v.b * d2.z);
    const float3 back = mimg3(make_float3(m.x - fresh.x, m.y - fresh.y, m.z - fresh.z), p);     // stored - new, minimum image: minus the true move
    m.x -= back.x; m.y -= back.y; m.z -= back.z;
    posq[v.site] = m;
}

__global__ void vsite_spread_kernel(uint32_t n, const VSite* __restrict__ vs, float4* __restrict__ force,
                                    const uint32_t* gate, uint32_t thr) {
    if (gate && *gate > thr) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const VSite v = vs[i];
    if (v.site == MDX_INVALID) return;
    const float4 fm = force[v.site];
    const float w0 = 1.0f - v.a - v.b;
    float4 f0 = force[v.p0], f1 = force[v.p1], f2 = force[v.p2];
    f0.x += w0 * fm.x; f0.y += w0 * fm.y; f0.z += w0 * fm.z;
    f1.x += v.a * fm.x; f1.y += v.a * fm.y; f1.z += v.a * fm.z;
    f2.x += v.b * fm.x; f2.y += v.b * fm.y; f2.z += v.b * fm.z;
    force[v.p0] = f0; force[v.p1] = f1; force[v.p2] = f2;
    force[v.site] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// On a decomposed handle (slot_flags given) a cluster is solved by the rank that OWNS it (ownership goes by cluster, so
// its first atom decides); everywhere else - absent, or present as ghosts whose positions arrive by halo message - the
// record is emptied.
// leaders (with a table): the slot of a cluster's first atom is marked in its tile's word - the clusters are then laid out in slot
// order by group_place_kernel, so that the threads of a wave of the solvers touch neighbouring slots.  (In caller order -
// waters as the host numbered them - every 16-byte access of the solvers was a cache line of its own: SETTLE at 1 M sites moved
// 3.6 x its algorithmic bytes and ran at 13 % of the HBM roofline.)
__global__ void remap_groups_kernel(uint32_t n, const ConsGroup* __restrict__ go, const uint32_t* __restrict__ slot_of,
                                    ConsGroup* __restrict__ gs, uint32_t* err, const uint8_t* __restrict__ slot_flags,
                                    unsigned long long* __restrict__ leaders, const GroupSite* __restrict__ so = nullptr,
                                    GroupSite* __restrict__ ss = nullptr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ConsGroup g = go[i];
    if (so) {      // the cluster's virtual site travels with it
        GroupSite v = so[i];
        if (v.on) { v.site = slot_of[v.site]; if (v.site == MDX_INVALID) v.on = 0; }
        ss[i] = v;
    }
    if (slot_flags) {
        const uint32_t s0 = slot_of[g.atom[0]];
        if (s0 == MDX_INVALID || !(slot_flags[s0] & 2u)) { g.natoms = 0; g.ncons = 0; gs[i] = g; if (so) ss[i].on = 0; return; }     // (not solved here: nor is its site placed here)
    }
    for (uint32_t k = 0; k < g.natoms; ++k) {
        const uint32_t s = slot_of[g.atom[k]];
        if (s == MDX_INVALID) atomicOr(err, 8u);
        g.atom[k] = s;
    }
    gs[i] = g;
    if (leaders && g.natoms && g.atom[0] != MDX_INVALID) atomicOr(leaders + (g.atom[0] >> 6), 1ull << (g.atom[0] & 63u));
}
__global__ void group_count_kernel(uint32_t nt, const unsigned long long* __restrict__ leaders, uint32_t* __restrict__ cnt) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= nt) cnt[t] = t < nt ? (uint32_t)__popcll(leaders[t]) : 0u;      // (the trailing element: the scan leaves the total there)
}
__global__ void group_place_kernel(uint32_t n, const ConsGroup* __restrict__ tmp, const unsigned long long* __restrict__ leaders,
                                   const uint32_t* __restrict__ off, ConsGroup* __restrict__ gs,
                                   const GroupSite* __restrict__ stmp = nullptr, GroupSite* __restrict__ ss = nullptr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ConsGroup g = tmp[i];
    if (!g.natoms || g.atom[0] == MDX_INVALID) return;      // not solved here: beyond the count the solvers read
    const uint32_t s0 = g.atom[0], t = s0 >> 6;
    const uint32_t at = off[t] + (uint32_t)__popcll(leaders[t] & ((1ull << (s0 & 63u)) - 1ull));
    gs[at] = g;
    if (stmp) ss[at] = stmp[i];
}

__global__ void remap_vsites_kernel(uint32_t n, const VSite* __restrict__ vo, const uint32_t* __restrict__ slot_of,
                                    VSite* __restrict__ vs, uint32_t* err, const uint8_t* __restrict__ slot_flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    VSite v = vo[i];
    if (slot_flags) {   // decomposed handle: the owner of the family constructs the site and spreads its force
        const uint32_t s0 = slot_of[v.p0];
        if (s0 == MDX_INVALID || !(slot_flags[s0] & 2u)) { v.site = MDX_INVALID; vs[i] = v; return; }
    }
    v.site = slot_of[v.site]; v.p0 = slot_of[v.p0]; v.p1 = slot_of[v.p1]; v.p2 = slot_of[v.p2];
    if (v.site == MDX_INVALID || v.p0 == MDX_INVALID || v.p1 == MDX_INVALID || v.p2 == MDX_INVALID) atomicOr(err, 8u);
    vs[i] = v;
}

// ---- host ---------------------------------------------------------------------------------------------
static ConsParams cons_params(const mdx_handle* h) {
    ConsParams p{};
    for (int d = 0; d < 3; ++d) {
        p.box[d] = h->per[d] ? (h->box_hi[d] - h->box_lo[d]) : 0.f;
        p.inv_box[d] = h->per[d] ? 1.0f / p.box[d] : 0.f;
    }
    p.tol = h->cfg.constraint_tol > 0.f ? h->cfg.constraint_tol : 1e-5f;
    p.max_iter = h->cfg.constraint_max_iter ? (int)h->cfg.constraint_max_iter : 64;
    static const bool settle_env = [] { const char* e = std::getenv("MDX_SETTLE"); return !(e && e[0] == '0'); }();   // A/B knob
    p.settle = settle_env ? 1 : 0;
    return p;
}

int mdx_build_constraints(mdx_handle* h, const mdx_system* s) {
    const uint32_t N = s->n_atoms;
    h->n_cons = s->n_constraints; h->n_vsites = s->n_vsites; h->n_groups = 0; h->n_star5 = 0; h->h_star5.clear();
    hipStream_t st = h->stream;
    if (s->n_constraints) {
        if (!s->constraint_idx || !s->constraint_len) FAIL(MDX_EPARAM, "missing constraint arrays");
        std::vector<uint32_t> parent(N);
        std::iota(parent.begin(), parent.end(), 0u);
        auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        for (uint32_t c = 0; c < s->n_constraints; ++c) {
            const uint32_t a = s->constraint_idx[2 * c], b = s->constraint_idx[2 * c + 1];
            if (a >= N || b >= N || a == b) FAIL(MDX_EPARAM, "bad constraint atom index");
            if (!(s->constraint_len[c] > 0.f)) FAIL(MDX_EPARAM, "constraint length must be positive");
            const bool fa = h->flags[a] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST), fb = h->flags[b] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST);
            if (fa && fb) FAIL(MDX_EPARAM, "constraint between two immobile atoms");
            parent[find(a)] = find(b);
        }
        // cluster sizes; a cluster of five atoms must be a star of four bonds (X-H4) and gets its own record
        std::vector<uint32_t> csize(N, 0);
        for (uint32_t i = 0; i < N; ++i) csize[find(i)]++;
        std::vector<int> star_of(N, -1);
        std::vector<ConsStar5> stars;
        for (uint32_t c = 0; c < s->n_constraints; ++c) {
            const uint32_t a = s->constraint_idx[2 * c], b = s->constraint_idx[2 * c + 1];
            const uint32_t root = find(a);
            if (csize[root] > 5) FAIL(MDX_EPARAM, "constraint cluster larger than 5 atoms (only rigid waters and heavy atoms with up to four hydrogens)");
            if (csize[root] != 5) continue;
            if (star_of[root] < 0) {
                star_of[root] = (int)stars.size();
                ConsStar5 st{};
                for (int k = 0; k < 5; ++k) st.atom[k] = MDX_INVALID;
                st.pad[0] = 0;      // bonds filed so far
                stars.push_back(st);
            }
            ConsStar5& st = stars[star_of[root]];
            const uint32_t nb = st.pad[0];
            if (nb == 0) { st.atom[0] = a; st.atom[1] = b; st.len[0] = s->constraint_len[c]; st.pad[0] = 1; continue; }
            if (nb == 1 && st.atom[0] != a && st.atom[0] != b) {      // the centre is the atom the first two bonds share
                if (st.atom[1] == a || st.atom[1] == b) std::swap(st.atom[0], st.atom[1]);
            }
            const uint32_t leaf = st.atom[0] == a ? b : (st.atom[0] == b ? a : MDX_INVALID);
            bool dup = false;
            for (uint32_t k = 1; k <= nb; ++k) dup = dup || st.atom[k] == leaf;
            if (leaf == MDX_INVALID || nb >= 4 || dup)
                FAIL(MDX_EPARAM, "a constraint cluster of five atoms must be a heavy atom with four constrained hydrogens (four bonds from one centre)");
            st.atom[nb + 1] = leaf; st.len[nb] = s->constraint_len[c]; st.pad[0] = nb + 1;
        }
        for (ConsStar5& st : stars) {
            if (st.pad[0] != 4) FAIL(MDX_EPARAM, "a constraint cluster of five atoms must be a heavy atom with four constrained hydrogens (four bonds from one centre)");
            st.pad[0] = 0;
        }
        h->n_star5 = (uint32_t)stars.size(); h->h_star5 = stars;
        std::vector<int> gid(N, -1);
        std::vector<ConsGroup> groups;
        for (uint32_t c = 0; c < s->n_constraints; ++c) {
            const uint32_t a = s->constraint_idx[2 * c], b = s->constraint_idx[2 * c + 1];
            const uint32_t root = find(a);
            if (csize[root] == 5) continue;
            if (gid[root] < 0) {
                gid[root] = (int)groups.size();
                ConsGroup g{};
                for (int k = 0; k < 4; ++k) g.atom[k] = MDX_INVALID;
                groups.push_back(g);
            }
            ConsGroup& g = groups[gid[root]];
            auto local = [&](uint32_t atom) -> int {
                for (uint32_t k = 0; k < g.natoms; ++k) if (g.atom[k] == atom) return (int)k;
                if (g.natoms >= 4) return -1;
                g.atom[g.natoms] = atom;
                return (int)g.natoms++;
            };
            const int la = local(a), lb = local(b);
            if (la < 0 || lb < 0 || g.ncons >= 6)
                FAIL(MDX_EPARAM, "constraint cluster of up to 4 atoms with more than 6 constraints");
            g.ca[g.ncons] = (uint8_t)la; g.cb[g.ncons] = (uint8_t)lb; g.len[g.ncons] = s->constraint_len[c];
            g.ncons++;
        }
        h->n_groups = (uint32_t)groups.size();
        h->h_groups = groups;
        // every cluster a rigid three-site water in canonical order (legs from atom 0, equal masses on atoms 1 and 2, constraints
        // exactly (0,1), (0,2), (1,2)): the solvers then run their RIGID3 flavour
        h->cons_all_rigid3 = !groups.empty();
        h->n_wstep_groups = 0;
        for (ConsGroup& g : groups) {
            bool ok = g.natoms == 3 && g.ncons == 3;
            float l01 = 0.f, l02 = 0.f, l12 = 0.f;
            for (uint32_t c = 0; ok && c < 3; ++c) {
                const int a = std::min((int)g.ca[c], (int)g.cb[c]), b = std::max((int)g.ca[c], (int)g.cb[c]);
                if (a == 0 && b == 1) l01 = g.len[c]; else if (a == 0 && b == 2) l02 = g.len[c]; else if (a == 1 && b == 2) l12 = g.len[c]; else ok = false;
            }
            ok = ok && l01 > 0.f && l02 > 0.f && l12 > 0.f && std::fabs(l01 - l02) <= 1e-6f * l01 && l12 < 2.0f * l01;
            if (ok) {
                const uint32_t a0 = g.atom[0], a1 = g.atom[1], a2 = g.atom[2];
                const bool mob = !(h->flags[a0] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST)) && !(h->flags[a1] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST)) &&
                                 !(h->flags[a2] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
                ok = mob && s->mass[a1] == s->mass[a2] && s->mass[a0] > 0.f && s->mass[a1] > 0.f;
            }
            g.wstep = ok ? 1u : 0u;      // (such a water's whole step may go through water_step_kernel)
            if (ok) ++h->n_wstep_groups; else h->cons_all_rigid3 = false;
        }
        h->h_groups = groups;
        for (void** q : {(void**)&h->d.cons_o, (void**)&h->d.cons_s, (void**)&h->d.cons_vir, (void**)&h->d.star_o, (void**)&h->d.star_s})
            if (*q) { (void)hipFree(*q); *q = nullptr; }
        const size_t n_all = groups.size() + stars.size();
        HIP_TRY(hipMalloc((void**)&h->d.cons_o, std::max<size_t>(sizeof(ConsGroup) * groups.size(), 16)));
        HIP_TRY(hipMalloc((void**)&h->d.cons_s, std::max<size_t>(sizeof(ConsGroup) * groups.size(), 16)));
        HIP_TRY(hipMalloc((void**)&h->d.cons_vir, sizeof(float) * n_all));
        HIP_TRY(hipMemsetAsync(h->d.cons_vir, 0, sizeof(float) * n_all, st));
        if (!groups.empty()) HIP_TRY(hipMemcpyAsync(h->d.cons_o, groups.data(), sizeof(ConsGroup) * groups.size(), hipMemcpyHostToDevice, st));
        if (!stars.empty()) {
            HIP_TRY(hipMalloc((void**)&h->d.star_o, sizeof(ConsStar5) * stars.size()));
            HIP_TRY(hipMalloc((void**)&h->d.star_s, sizeof(ConsStar5) * stars.size()));
            HIP_TRY(hipMemcpyAsync(h->d.star_o, stars.data(), sizeof(ConsStar5) * stars.size(), hipMemcpyHostToDevice, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (s->n_vsites) {
        if (!s->vsite_idx || !s->vsite_w) FAIL(MDX_EPARAM, "missing virtual-site arrays");
        std::vector<VSite> vs(s->n_vsites);
        std::vector<uint8_t> used(N, 0);
        for (uint32_t i = 0; i < s->n_vsites; ++i) {
            VSite v{};
            v.site = s->vsite_idx[4 * i]; v.p0 = s->vsite_idx[4 * i + 1]; v.p1 = s->vsite_idx[4 * i + 2];
            v.p2 = s->vsite_idx[4 * i + 3]; v.a = s->vsite_w[2 * i]; v.b = s->vsite_w[2 * i + 1];
            // a site inside the triangle of its parents never moves further than they do (dual pair list)
            if (!(v.a >= 0.f && v.b >= 0.f && v.a + v.b <= 1.f)) h->vsites_convex = false;
            const uint32_t ids[4] = {v.site, v.p0, v.p1, v.p2};
            for (uint32_t id : ids) {
                if (id >= N) FAIL(MDX_EPARAM, "virtual-site atom index out of range");
                if (used[id]++) FAIL(MDX_EPARAM, "an atom may be the site or a parent of one virtual site only");
            }
            if (!(h->flags[v.site] & MDX_ATOM_STATIC))
                FAIL(MDX_EPARAM, "a virtual site must be flagged MDX_ATOM_STATIC (it is massless and never integrated)");
            vs[i] = v;
        }
        h->h_vsites = vs;
        if (h->d.vsite_o) (void)hipFree(h->d.vsite_o);
        if (h->d.vsite_s) (void)hipFree(h->d.vsite_s);
        HIP_TRY(hipMalloc((void**)&h->d.vsite_o, sizeof(VSite) * vs.size()));
        HIP_TRY(hipMalloc((void**)&h->d.vsite_s, sizeof(VSite) * vs.size()));
        HIP_TRY(hipMemcpyAsync(h->d.vsite_o, vs.data(), sizeof(VSite) * vs.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    {   // does every excluded pair lie inside one rigid three-site cluster (site included)?  (mdx_step: such corrections change nothing
        // the constraints do not undo, and the step loop of a rigid-water box leaves them to the force calls whose result is read)
        h->excl_inside_rigid = false;
        if (s->excl_offsets && s->excl_idx && h->cons_all_rigid3 && !h->h_groups.empty()) {
            std::vector<int> cl(N, -1);
            for (size_t gi = 0; gi < h->h_groups.size(); ++gi)
                for (uint32_t k = 0; k < h->h_groups[gi].natoms; ++k) cl[h->h_groups[gi].atom[k]] = (int)gi;
            bool ok = true;
            for (const VSite& v : h->h_vsites) {      // a site belongs to the cluster all three of its parents are members of
                if (cl[v.p0] >= 0 && cl[v.p0] == cl[v.p1] && cl[v.p0] == cl[v.p2]) cl[v.site] = cl[v.p0]; else ok = false;
            }
            for (uint32_t i = 0; i < N && ok; ++i)
                for (uint32_t k = s->excl_offsets[i]; k < s->excl_offsets[i + 1]; ++k) {
                    const uint32_t j = s->excl_idx[k];
                    if (cl[i] < 0 || cl[i] != cl[j]) { ok = false; break; }
                }
            h->excl_inside_rigid = ok;
        }
    }
    // Sites whose three parents are the members of ONE constraint cluster (rigid four-site water) are placed by that cluster's
    // position stage (GroupSite).  All of them or none: MDX_VSITE_IN_GROUPS=0 keeps the construct launch (A/B).
    h->vsites_in_groups = false; h->vsites_fresh = false;
    for (void** q : {(void**)&h->d.gsite_o, (void**)&h->d.gsite_s, (void**)&h->d.gsite_tmp}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    const char* const ve = std::getenv("MDX_VSITE_IN_GROUPS");
    if (h->n_vsites && h->n_groups && !(ve && ve[0] == '0')) {
        std::vector<int> grp(N, -1), loc(N, -1);
        for (size_t gi = 0; gi < h->h_groups.size(); ++gi)
            for (uint32_t k = 0; k < h->h_groups[gi].natoms; ++k) { grp[h->h_groups[gi].atom[k]] = (int)gi; loc[h->h_groups[gi].atom[k]] = (int)k; }
        std::vector<GroupSite> gsv(h->h_groups.size());
        for (auto& x : gsv) { x.site = MDX_INVALID; x.k0 = x.k1 = x.k2 = x.on = 0; x.a = x.b = 0.f; }
        size_t attached = 0;
        for (const VSite& v : h->h_vsites) {
            const int gi = grp[v.p0];
            if (gi < 0 || grp[v.p1] != gi || grp[v.p2] != gi || gsv[gi].on) continue;
            gsv[gi].site = v.site; gsv[gi].k0 = (uint8_t)loc[v.p0]; gsv[gi].k1 = (uint8_t)loc[v.p1]; gsv[gi].k2 = (uint8_t)loc[v.p2];
            gsv[gi].on = 1; gsv[gi].a = v.a; gsv[gi].b = v.b;
            ++attached;
        }
        if (attached == h->h_vsites.size()) {
            HIP_TRY(hipMalloc((void**)&h->d.gsite_o, sizeof(GroupSite) * gsv.size()));
            HIP_TRY(hipMalloc((void**)&h->d.gsite_s, sizeof(GroupSite) * gsv.size()));
            HIP_TRY(hipMemcpyAsync(h->d.gsite_o, gsv.data(), sizeof(GroupSite) * gsv.size(), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            h->vsites_in_groups = true;
        }
    }
    {   // what the one-pass water step covers: every site its cluster's?  every atom of the handle?
        std::vector<int> wcl(N, -1);
        for (size_t gi = 0; gi < h->h_groups.size(); ++gi)
            if (h->h_groups[gi].wstep) for (uint32_t k = 0; k < 3; ++k) wcl[h->h_groups[gi].atom[k]] = (int)gi;
        // (the kernel reaches a site through its cluster's GroupSite record: without those - MDX_VSITE_IN_GROUPS=0, or a site whose
        // parents are not one cluster - the sites keep their own construct / spread launches)
        h->wstep_sites_all = h->h_vsites.empty() || h->vsites_in_groups;
        for (const VSite& v : h->h_vsites)
            if (!(wcl[v.p0] >= 0 && wcl[v.p0] == wcl[v.p1] && wcl[v.p0] == wcl[v.p2])) h->wstep_sites_all = false;
        h->wstep_all = h->n_wstep_groups != 0 && h->n_wstep_groups == h->n_groups && h->n_star5 == 0 && h->wstep_sites_all &&
                       h->n_mobile == 3u * h->n_wstep_groups && N == h->n_mobile + h->n_vsites;
    }
    return MDX_OK;
}

// mixed systems: bit 4 of wstep_s[slot] <=> the slot is a member (or the site) of a rigid water that water_step_kernel steps
__global__ void mark_wstep_kernel(uint32_t n, const ConsGroup* __restrict__ gs, const GroupSite* __restrict__ ss, const uint32_t* __restrict__ n_dev,
                                  uint8_t* __restrict__ wstep_s) {
    if (n_dev) n = min(n, *n_dev);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ConsGroup g = gs[i];
    if (!g.wstep || g.natoms != 3) return;
    for (int k = 0; k < 3; ++k) if (g.atom[k] != MDX_INVALID) wstep_s[g.atom[k]] = 4u;
    if (ss) { const GroupSite v = ss[i]; if (v.on && v.site != MDX_INVALID) wstep_s[v.site] = 4u; }
}
static int mark_wstep_slots(mdx_handle* h) {
    DeviceState& d = h->d;
    if (!h->n_wstep_groups || h->wstep_all || h->dd || h->n_local != h->N) return MDX_OK;
    const uint32_t need = h->cap_tiles * MDX_TILE;
    if (d.cap_wstep < need) {
        if (d.wstep_s) (void)hipFree(d.wstep_s);
        d.wstep_s = nullptr; d.cap_wstep = 0;
        HIP_TRY(hipMalloc((void**)&d.wstep_s, need));
        d.cap_wstep = need;
    }
    HIP_TRY(hipMemsetAsync(d.wstep_s, 0, need, h->stream));
    hipLaunchKernelGGL(mark_wstep_kernel, dim3(div_up(h->n_groups, 256)), dim3(256), 0, h->stream, h->n_groups, d.cons_s,
                       (const GroupSite*)(h->vsites_in_groups ? d.gsite_s : nullptr), d.cons_n_dev, d.wstep_s);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

static int remap_constraints_impl(mdx_handle* h);
int mdx_remap_constraints(mdx_handle* h) {
    MDX_TRY(remap_constraints_impl(h));
    return mark_wstep_slots(h);
}
static int remap_constraints_impl(mdx_handle* h) {
    static const bool sort_env = [] { const char* e = std::getenv("MDX_CONS_SORT"); return !(e && e[0] == '0'); }();   // A/B knob
    DeviceState& d = h->d;
    const uint8_t* const sf = (h->dd || h->n_local != h->N) ? d.slot_flags : nullptr;
    // (below 32 k clusters the solvers are latency-bound and the four extra launches per rebuild cost more than they save;
    // MDX_CONS_SORT_MIN: another threshold - the tests use 1 to take small systems through this path)
    const char* const me = std::getenv("MDX_CONS_SORT_MIN");
    const uint32_t sort_min = me ? (uint32_t)std::max(1, std::atoi(me)) : 32768u;
    if (h->n_groups && sort_env && h->n_groups >= sort_min && h->cap_tiles) {
        // clusters in slot order (every tile word of leaders is at most 64 clusters; a slot leads at most one cluster)
        const uint32_t nt = h->cap_tiles;
        if (d.cons_cap_tiles < nt || !d.cons_tmp) {
            for (void** q : {(void**)&d.cons_tmp, (void**)&d.cons_mask, (void**)&d.cons_cnt, (void**)&d.cons_off})
                if (*q) { (void)hipFree(*q); *q = nullptr; }
            HIP_TRY(hipMalloc((void**)&d.cons_tmp, sizeof(ConsGroup) * h->n_groups));
            HIP_TRY(hipMalloc((void**)&d.cons_mask, sizeof(unsigned long long) * nt));
            HIP_TRY(hipMalloc((void**)&d.cons_cnt, sizeof(uint32_t) * ((size_t)nt + 1)));
            HIP_TRY(hipMalloc((void**)&d.cons_off, sizeof(uint32_t) * ((size_t)nt + 1 + nt / 2048 + 64)));
            d.cons_cap_tiles = nt;
        }
        HIP_TRY(hipMemsetAsync(d.cons_mask, 0, sizeof(unsigned long long) * nt, h->stream));
        if (d.gsite_o && !d.gsite_tmp) HIP_TRY(hipMalloc((void**)&d.gsite_tmp, sizeof(GroupSite) * h->n_groups));
        hipLaunchKernelGGL(remap_groups_kernel, dim3(div_up(h->n_groups, 256)), dim3(256), 0, h->stream, h->n_groups,
                           d.cons_o, d.slot_of, d.cons_tmp, d.flags_dev, sf, d.cons_mask, (const GroupSite*)d.gsite_o, d.gsite_tmp);
        hipLaunchKernelGGL(group_count_kernel, dim3(div_up(nt + 1, 256)), dim3(256), 0, h->stream, nt, d.cons_mask, d.cons_cnt);
        MDX_TRY(mdx_exclusive_scan_u32_ex(h, d.cons_cnt, d.cons_off, nt + 1, d.cons_off + nt + 1));
        hipLaunchKernelGGL(group_place_kernel, dim3(div_up(h->n_groups, 256)), dim3(256), 0, h->stream, h->n_groups,
                           d.cons_tmp, d.cons_mask, d.cons_off, d.cons_s, (const GroupSite*)(d.gsite_o ? d.gsite_tmp : nullptr), d.gsite_s);
        d.cons_n_dev = d.cons_off + nt;
    } else if (h->n_groups) {
        hipLaunchKernelGGL(remap_groups_kernel, dim3(div_up(h->n_groups, 256)), dim3(256), 0, h->stream, h->n_groups,
                           d.cons_o, d.slot_of, d.cons_s, d.flags_dev, sf, (unsigned long long*)nullptr, (const GroupSite*)d.gsite_o, d.gsite_s);
        d.cons_n_dev = nullptr;
    }
    if (h->n_star5)
        hipLaunchKernelGGL(remap_star5_kernel, dim3(div_up(h->n_star5, 256)), dim3(256), 0, h->stream, h->n_star5, d.star_o, d.slot_of, d.star_s, d.flags_dev, sf);
    if (h->n_vsites)
        hipLaunchKernelGGL(remap_vsites_kernel, dim3(div_up(h->n_vsites, 256)), dim3(256), 0, h->stream, h->n_vsites,
                           h->d.vsite_o, h->d.slot_of, h->d.vsite_s, h->d.flags_dev, (h->dd || h->n_local != h->N) ? h->d.slot_flags : nullptr);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_constrain_positions(mdx_handle* h, float dt, const uint32_t* d_gate, uint32_t* d_disp_out, uint32_t thr,
                                   uint32_t* d_prune_out, bool skip_wstep) {
    if (!mdx_has_constraints(h)) return MDX_OK;
    if (!h->dual_on) d_prune_out = nullptr;
    ConsParams cp = cons_params(h);
    cp.vir_scale = h->cons_full_kick ? 1.0f : 2.0f;
    if (h->n_star5)
        hipLaunchKernelGGL(constrain_positions_star5_kernel, dim3(div_up(h->n_star5, 128)), dim3(128), 0, h->stream, h->n_star5, h->d.star_s, h->d.posq,
                           h->d.vel, h->d.ref, dt, cp, dt != 0.f ? h->d.cons_vir + h->n_groups : nullptr, d_gate, d_disp_out, thr, d_prune_out,
                           0.5f * h->inner_skin * (1.0f - 1.0e-4f));
    if (!h->n_groups) { HIP_TRY(hipGetLastError()); return MDX_OK; }
    // (MDX_CONS_RIGID3=0: the general flavour for every handle, A/B)
    static const bool rigid3_env = [] { const char* e = std::getenv("MDX_CONS_RIGID3"); return !(e && e[0] == '0'); }();
    auto kern = (h->cons_all_rigid3 && rigid3_env) ? constrain_positions_kernel<true> : constrain_positions_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(div_up(h->n_groups, 128)), dim3(128), 0, h->stream, h->n_groups,
                       h->d.cons_s, h->d.posq, h->d.vel, h->d.ref, dt, cp,
                       dt != 0.f ? h->d.cons_vir : nullptr,   // a dt = 0 projection (new coordinates, rescaled box) keeps the last step's virial
                       d_gate, d_disp_out, thr, d_prune_out, 0.5f * h->inner_skin * (1.0f - 1.0e-4f), h->d.cons_n_dev,
                       (const GroupSite*)(h->vsites_in_groups ? h->d.gsite_s : nullptr), skip_wstep ? 1u : 0u);
    if (h->vsites_in_groups) h->vsites_fresh = true;
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

__global__ __launch_bounds__(256) void constraint_virial_kernel(uint32_t n, const float* __restrict__ cons_vir, double* energy,
                                                                const uint32_t* __restrict__ n_dev) {
    if (n_dev) n = min(n, *n_dev);
    double w = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) w += (double)cons_vir[i];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) w += __shfl_xor(w, m);
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(&energy[EN_VIRIAL], w);
}

int mdx_launch_constraint_virial(mdx_handle* h) {
    if (!mdx_has_constraints(h) || !h->d.cons_vir) return MDX_OK;
    if (h->n_star5)
        hipLaunchKernelGGL(constraint_virial_kernel, dim3(std::min<uint32_t>(div_up(h->n_star5, 256), 256u)), dim3(256), 0, h->stream,
                           h->n_star5, h->d.cons_vir + h->n_groups, h->d.energy, (const uint32_t*)nullptr);
    if (!h->n_groups) { HIP_TRY(hipGetLastError()); return MDX_OK; }
    hipLaunchKernelGGL(constraint_virial_kernel, dim3(std::min<uint32_t>(div_up(h->n_groups, 256), 256u)), dim3(256), 0, h->stream,
                       h->n_groups, h->d.cons_vir, h->d.energy, h->d.cons_n_dev);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}


// ---- one pass per step for boxes of rigid water (round 6) -----------------------------------------------------------------------
// The reference's default operating point is a box of rigid four-site OPC water at dt = 2 fs (/root/reference src/prefs/mod.rs:203,
// src/ui/panels/md.rs:362-371).  Per step such a handle ran vsite_spread_kernel (M's force onto O, H, H: 17 us at 1 M sites),
// integrate_kernel (kick + drift: 25 us) and constrain_positions_kernel (SETTLE + M placed: 35 us) - three passes over the same
// rows.  Where EVERY mobile atom is a member of a rigid three-site cluster and every virtual site is its cluster's (host-checked:
// mdx_water_step_ok) a lane takes one water through all of it: force of the site handed to its parents, kick, drift, SETTLE in
// closed form on the old / new triangle, velocities corrected, site placed, forces cleared for the pair kernel's atomics, rebuild
// and pruning triggers raised - every row read once and written once.  KICK 0: opening half kick (first step of a chunk), 1: full
// kick (closing + opening).  The arithmetic of the pieces is the separate kernels' (same SETTLE, same image convention); the old
// positions are the stored ones instead of x' - dt v'.  MDX_WATER_STEP=0: the three launches (A/B).
template <int KICK>
__global__ __launch_bounds__(128) void water_step_kernel(uint32_t n_groups, const ConsGroup* __restrict__ groups, const GroupSite* __restrict__ gsite,
                                                         float4* __restrict__ posq, float4* __restrict__ vel, float4* __restrict__ force,
                                                         float4* __restrict__ ref, float dt, ConsParams p, float* __restrict__ cons_vir,
                                                         const uint32_t* gate_in, uint32_t* disp_out, uint32_t thr, uint32_t* prune_out,
                                                         float path_thr, const uint32_t* __restrict__ n_dev, uint32_t spread, uint32_t zero) {
    const uint32_t gate = gate_in ? *gate_in : 0u;
    if (gate > thr) {      // list already stale: stay a no-op, keep the flag raised (integrate_kernel's rule)
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(disp_out, gate);
        return;
    }
    if (n_dev) n_groups = min(n_groups, *n_dev);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float d2max = 0.f, pmax = 0.f;
    if (g < n_groups) {
        const ConsGroup cg = groups[g];
        if (cg.natoms == 3 && cg.wstep) {
            GroupSite gv{}; gv.site = MDX_INVALID;
            if (gsite) gv = gsite[g];
            const bool have_site = gv.on && gv.site != MDX_INVALID;
            float4 pk[3], vk[3], fk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { pk[k] = posq[cg.atom[k]]; vk[k] = vel[cg.atom[k]]; fk[k] = force[cg.atom[k]]; }
            if (have_site && spread) {      // the site's force belongs to its parents (vsite_spread_kernel's weights)
                const float4 fm = force[gv.site];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float wk = (k == (int)gv.k0 ? 1.0f - gv.a - gv.b : 0.f) + (k == (int)gv.k1 ? gv.a : 0.f) + (k == (int)gv.k2 ? gv.b : 0.f);
                    fk[k].x += wk * fm.x; fk[k].y += wk * fm.y; fk[k].z += wk * fm.z;
                }
            }
            float3 xo[4], xn[4], xs[4];
            float im[4];
            xo[3] = xn[3] = xs[3] = make_float3(0, 0, 0); im[3] = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float kdt = (KICK == 1 ? dt : 0.5f * dt) * vk[k].w;
                vk[k].x += kdt * fk[k].x; vk[k].y += kdt * fk[k].y; vk[k].z += kdt * fk[k].z;
                xo[k] = mimg3(make_float3(pk[k].x - pk[0].x, pk[k].y - pk[0].y, pk[k].z - pk[0].z), p);
                xn[k] = make_float3(xo[k].x + dt * vk[k].x, xo[k].y + dt * vk[k].y, xo[k].z + dt * vk[k].z);
                xs[k] = xn[k];
                im[k] = vk[k].w;
            }
            float wc = 0.f;
            bool settled = false;
            const Triangle tri = rigid_triangle(cg, im);
            if (p.settle && tri.ok && settle_positions(xo, xn, im, tri)) {
                settled = true;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    wc += ((xn[k].x - xs[k].x) * xo[k].x + (xn[k].y - xs[k].y) * xo[k].y + (xn[k].z - xs[k].z) * xo[k].z) / im[k];
            }
            if (!settled) {      // degenerate step or SETTLE switched off: the Gauss-Seidel sweep over the three pairs
                const float Lq[3] = {tri.l01, tri.l01, tri.l12};
                for (int it = 0; it < p.max_iter; ++it) {
                    bool done = true;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int a = q == 2 ? 1 : 0, b = q == 0 ? 1 : 2;
                        if (!(Lq[q] > 0.f)) continue;
                        const float3 s = make_float3(xn[a].x - xn[b].x, xn[a].y - xn[b].y, xn[a].z - xn[b].z);
                        const float l2 = Lq[q] * Lq[q];
                        const float diff = l2 - (s.x * s.x + s.y * s.y + s.z * s.z);
                        if (fabsf(diff) > 2.0f * p.tol * l2) {
                            done = false;
                            const float3 r = make_float3(xo[a].x - xo[b].x, xo[a].y - xo[b].y, xo[a].z - xo[b].z);
                            const float sr = s.x * r.x + s.y * r.y + s.z * r.z;
                            const float gk = diff / (2.0f * sr * (im[a] + im[b]));
                            wc += gk * (r.x * r.x + r.y * r.y + r.z * r.z);
                            xn[a].x += gk * im[a] * r.x; xn[a].y += gk * im[a] * r.y; xn[a].z += gk * im[a] * r.z;
                            xn[b].x -= gk * im[b] * r.x; xn[b].y -= gk * im[b] * r.y; xn[b].z -= gk * im[b] * r.z;
                        }
                    }
                    if (done) break;
                }
            }
            const float idt = 1.0f / dt;
            if (cons_vir) cons_vir[g] = p.vir_scale * wc * idt * idt;
            float4 p0n = pk[0];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float4 pn = pk[k];
                const float mx = xn[k].x - xo[k].x, my = xn[k].y - xo[k].y, mz = xn[k].z - xo[k].z;      // this step's displacement of the atom
                pn.x += mx; pn.y += my; pn.z += mz;
                vk[k].x += (xn[k].x - xs[k].x) * idt; vk[k].y += (xn[k].y - xs[k].y) * idt; vk[k].z += (xn[k].z - xs[k].z) * idt;
                posq[cg.atom[k]] = pn; vel[cg.atom[k]] = vk[k];
                if (zero) force[cg.atom[k]] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k == 0) p0n = pn;
                const float4 r = ref[cg.atom[k]];
                const float ex = pn.x - r.x, ey = pn.y - r.y, ez = pn.z - r.z;
                d2max = fmaxf(d2max, ex * ex + ey * ey + ez * ez);
                if (prune_out) {
                    const float w = r.w + sqrtf(mx * mx + my * my + mz * mz);
                    ref[cg.atom[k]].w = w;
                    pmax = fmaxf(pmax, w);
                }
            }
            if (have_site) {
                // r_p0 + a (r_p1 - r_p0) + b (r_p2 - r_p0) from the constrained positions in registers, at the periodic image nearest to
                // where the site is stored (constrain_positions_kernel's rule)
                float3 fresh = make_float3(p0n.x - xn[0].x, p0n.y - xn[0].y, p0n.z - xn[0].z);      // the frame's origin (the old position of atom 0)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float wk = (k == (int)gv.k0 ? 1.0f - gv.a - gv.b : 0.f) + (k == (int)gv.k1 ? gv.a : 0.f) + (k == (int)gv.k2 ? gv.b : 0.f);
                    fresh.x += wk * xn[k].x; fresh.y += wk * xn[k].y; fresh.z += wk * xn[k].z;
                }
                float4 m = posq[gv.site];
                const float3 back = mimg3(make_float3(m.x - fresh.x, m.y - fresh.y, m.z - fresh.z), p);
                m.x -= back.x; m.y -= back.y; m.z -= back.z;
                posq[gv.site] = m;
                if (zero || spread) force[gv.site] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (!(d2max < 1.0e30f)) d2max = 3.0e38f;      // NaN / inf -> huge: forces a stop
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) d2max = fmaxf(d2max, __shfl_xor(d2max, m));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(d2max) > thr) atomicMax(disp_out, __float_as_uint(d2max));
    if (prune_out) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, m));
        if ((threadIdx.x & 63) == 0 && !(pmax <= path_thr)) *prune_out = 1u;
    }
}

// Is this handle a box of rigid water the one-pass step applies to?  (every mobile atom a member of a rigid three-site cluster, every
// other atom the virtual site of such a cluster, velocity Verlet, one device)
bool mdx_water_step_ok(const mdx_handle* h) {
    static const bool on = [] { const char* e = std::getenv("MDX_WATER_STEP"); return !(e && e[0] == '0'); }();
    static const bool mixed_on = [] { const char* e = std::getenv("MDX_WATER_STEP_MIXED"); return !(e && e[0] == '0'); }();
    return on && h->integrator == MDX_INTEGRATOR_VERLET_VELOCITY && h->n_wstep_groups != 0 && (h->wstep_all || mixed_on) &&
           !h->dd && h->n_local == h->N && mdx_nb_variant(h) >= 2;
}
// ... with other mobile atoms / clusters beside the waters (a solute in rigid water - what the reference's users run): those keep
// integrate_kernel (which then skips the waters' slots: d.wstep_s) and the cluster solvers (which skip the waters' records)
bool mdx_water_step_mixed(const mdx_handle* h) { return !h->wstep_all; }

// mode 0: opening half kick + drift, 1: full kick + drift (mdx_launch_integrate's modes); then SETTLE and the site, all in one launch
int mdx_launch_water_step(mdx_handle* h, int mode, float dt, const uint32_t* d_gate, uint32_t* d_disp_out, uint32_t thr, uint32_t* d_prune_out) {
    if (!h->dual_on) d_prune_out = nullptr;
    ConsParams cp = cons_params(h);
    cp.vir_scale = mode == 1 ? 1.0f : 2.0f;      // (mdx_launch_constrain_positions: cons_full_kick)
    const uint32_t zero = mdx_nb_half(h) ? 1u : 0u, spread = (h->n_vsites && h->vsite_spread_pending) ? 1u : 0u;
    h->vsite_spread_pending = false;
    const GroupSite* const gs = h->vsites_in_groups ? (const GroupSite*)h->d.gsite_s : nullptr;
    const dim3 g(div_up(h->n_groups, 128)), b(128);
    if (!h->wstep_all && !h->d.wstep_s) FAIL(MDX_EDEVICE, "internal: the slots of the rigid waters are not marked");
    mdx_prof_begin(h, 2);
    if (mode == 1)
        hipLaunchKernelGGL(water_step_kernel<1>, g, b, 0, h->stream, h->n_groups, h->d.cons_s, gs, h->d.posq, h->d.vel, h->d.force, h->d.ref, dt, cp,
                           h->d.cons_vir, d_gate, d_disp_out, thr, d_prune_out, 0.5f * h->inner_skin * (1.0f - 1.0e-4f), h->d.cons_n_dev, spread, zero);
    else
        hipLaunchKernelGGL(water_step_kernel<0>, g, b, 0, h->stream, h->n_groups, h->d.cons_s, gs, h->d.posq, h->d.vel, h->d.force, h->d.ref, dt, cp,
                           h->d.cons_vir, d_gate, d_disp_out, thr, d_prune_out, 0.5f * h->inner_skin * (1.0f - 1.0e-4f), h->d.cons_n_dev, spread, zero);
    mdx_prof_end(h);
    ++h->water_step_launches;
    if (!h->wstep_all) ++h->water_step_mixed_launches;
    if (zero && h->wstep_all) h->force_zeroed = true;      // (mixed systems: integrate_kernel clears the other slots and says so)
    if (h->vsites_in_groups && h->wstep_sites_all) h->vsites_fresh = true;
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_constrain_velocities(mdx_handle* h, const uint32_t* d_gate, uint32_t thr) {
    if (!mdx_has_constraints(h)) return MDX_OK;
    if (h->n_star5)
        hipLaunchKernelGGL(constrain_velocities_star5_kernel, dim3(div_up(h->n_star5, 128)), dim3(128), 0, h->stream, h->n_star5, h->d.star_s,
                           h->d.posq, h->d.vel, cons_params(h), d_gate, thr);
    if (!h->n_groups) { HIP_TRY(hipGetLastError()); return MDX_OK; }
    static const bool rigid3_env = [] { const char* e = std::getenv("MDX_CONS_RIGID3"); return !(e && e[0] == '0'); }();
    auto kern = (h->cons_all_rigid3 && rigid3_env) ? constrain_velocities_kernel<true> : constrain_velocities_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(div_up(h->n_groups, 128)), dim3(128), 0, h->stream, h->n_groups,
                       h->d.cons_s, h->d.posq, h->d.vel, cons_params(h), d_gate, thr, h->d.cons_n_dev);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_vsite_construct(mdx_handle* h, const uint32_t* d_gate, uint32_t thr) {
    if (!h->n_vsites) return MDX_OK;
    // every site was placed by the position stage of its cluster's constraint solver, and nothing has moved a parent since
    if (h->vsites_fresh) { h->vsites_fresh = false; return MDX_OK; }
    hipLaunchKernelGGL(vsite_construct_kernel, dim3(div_up(h->n_vsites, 256)), dim3(256), 0, h->stream, h->n_vsites,
                       h->d.vsite_s, h->d.posq, cons_params(h), d_gate, thr);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_vsite_spread(mdx_handle* h, const uint32_t* d_gate, uint32_t thr) {
    if (!h->n_vsites) return MDX_OK;
    // step loop of a rigid-water box: the next step's one-pass kernel hands the sites' forces to their parents itself
    if (h->vsite_spread_deferred) { h->vsite_spread_pending = true; return MDX_OK; }
    h->vsite_spread_pending = false;
    hipLaunchKernelGGL(vsite_spread_kernel, dim3(div_up(h->n_vsites, 256)), dim3(256), 0, h->stream, h->n_vsites,
                       h->d.vsite_s, h->d.force, d_gate, thr);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

// ---- HydrogenConstraint as the reference's UI states it (include/mdx.h) ------------------------------------------------
extern "C" int mdx_set_hydrogen_constraint(mdx_handle* h, int kind, uint32_t lincs_order, uint32_t lincs_iter, float shake_tolerance) {
    if (!h) { mdx_set_error("null handle"); return MDX_EPARAM; }
    if (kind == MDX_HC_FLEXIBLE) {
        if (mdx_has_constraints(h)) { mdx_set_error("HydrogenConstraint::Flexible on a system created with constraints: build it without them"); return MDX_EPARAM; }
        h->hc_kind = kind; return MDX_OK;
    }
    if (kind == MDX_HC_SHAKE) {
        if (std::isfinite(shake_tolerance) && shake_tolerance > 0.f) h->cfg.constraint_tol = shake_tolerance;
        h->hc_kind = kind; return MDX_OK;
    }
    if (kind != MDX_HC_LINEAR) { mdx_set_error("unknown hydrogen-constraint kind"); return MDX_EPARAM; }
    if (lincs_order == 0 || lincs_order > 16 || lincs_iter == 0 || lincs_iter > 16) { mdx_set_error("LINCS order / iter out of range (1..16)"); return MDX_EPARAM; }
    // LINCS leaves ~ (coupling)^order per pass; the converged solver is asked for no less than that, and never for less than it
    // was configured to deliver
    const float lincs_tol = std::pow(10.0f, -(0.5f * (float)lincs_order + (float)lincs_iter + 1.0f));
    const float cur = h->cfg.constraint_tol > 0.f ? h->cfg.constraint_tol : 1e-5f;
    h->cfg.constraint_tol = std::min(cur, lincs_tol);
    h->hc_kind = kind; h->hc_order = lincs_order; h->hc_iter = lincs_iter;
    return MDX_OK;
}

extern "C" const char* mdx_constraint_description(mdx_handle* h) {
    if (!h) return "";
    char buf[384];
    const float tol = h->cfg.constraint_tol > 0.f ? h->cfg.constraint_tol : 1e-5f;
    if (!mdx_has_constraints(h)) std::snprintf(buf, sizeof(buf), "no constraints (flexible)");
    else if (h->hc_kind == MDX_HC_LINEAR)
        std::snprintf(buf, sizeof(buf), "Linear{order %u, iter %u} mapped onto the converged cluster solver: %u clusters, SHAKE + RATTLE in registers to a relative "
                      "tolerance of %.1e (rigid three-site waters in closed form, SETTLE); LINCS at these settings would leave ~%.0e",
                      h->hc_order, h->hc_iter, h->n_groups + h->n_star5, (double)tol, std::pow(10.0, -(0.5 * h->hc_order + h->hc_iter + 1.0)));
    else std::snprintf(buf, sizeof(buf), "Shake: %u clusters, SHAKE + RATTLE in registers to a relative tolerance of %.1e (rigid three-site waters in closed form, SETTLE)",
                       h->n_groups + h->n_star5, (double)tol);
    h->hc_text = buf;
    return h->hc_text.c_str();
}
