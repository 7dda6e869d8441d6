/*
 * mdx_oracle.c — CPU restatement (plain C, fp64) of the MD force / integrate path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (molchanica_amd/, libmdx.so) links,
 * imports or executes this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may.
 *
 * PIN.  The engine the reference calls is the third-party crate `dynamics = "0.2.2"` (Cargo.toml:25; + ewald 0.1.15,
 * lin_alg 1.4.3, bio_files 0.5.3): absent from /root/reference, unbuildable here (no Rust toolchain), and the reference
 * holds no test or golden vector on this path (src/tests.rs:3-4 is empty).  What the tree DOES hold in native code is
 * pinned by running it: the reference's own CUDA kernels (src/cuda/cuda.cu + util.cu: lj_force_kernel,
 * coulomb_force_kernel, lj_V_kernel, min_image) compile with hipcc as they lie into oracle/_ref/libref_cuda.so
 * (oracle/Makefile target `ref`), run on the MI355X, and
 *   - this file's LJ 12-6 force / energy, tgt - src direction and Coulomb form with softening reproduce them on random
 *     sets (tests/test_gpu_reference_kernels.py);
 *   - its PRODUCTION-PATH sums do: for 300 atoms of dhfr23k the nonbonded force equals lj_force_kernel + k_e *
 *     coulomb_force_kernel of the reference on the atom's pre-imaged, cutoff-filtered, exclusion-filtered source set
 *     (tests/test_reference_pin.py, tests/ref_cases.py) - the cutoff filter, periodic images, exclusion and 1-4 removal and
 *     the Lorentz-Berthelot tables all enter through that set;
 *   - its canonical minimum image takes the reference's image on every tie (d = +-L/2, +-3L/2, ...) and differs from the
 *     reference's compiled value only by the one rounding fp-contraction saves (tests/test_reference_pin.py);
 *   - the kernels' outputs are recorded in tests/golden/ref_pair_kernels.npz (tests/golden/make_ref_pair_kernels.py), so the
 *     pin survives a checkout without /root/reference.
 * STILL UNPINNED BY THE REFERENCE (decided by the absent crate alone; held by analytic known-answer tests, finite
 * differences, conservation laws and literature values instead - tests/test_oracle_*.py, tests/test_gpu_physical_pins.py):
 * the value of k_e, 1-4 scaling factors, the Coulomb treatment AT the cutoff (energy shift / reaction field / Ewald), the
 * bonded conventions, integrator order, constraints, thermostats, SPME.
 *
 * What it follows in the reference tree (paths relative to /root/reference):
 *   - LJ 12-6 force/energy form and the `tgt - src` direction convention: src/cuda/util.cu:92-140
 *   - Coulomb form dir*q_s*q_t/(r^2 + soft):                           src/cuda/util.cu:53-63
 *   - minimum image d - rint(d/L)*L per axis, guard extent>0:          src/cuda/util.cu:65-71,
 *                                                                       src/md/mod.rs:278-296
 *   - parameter record shapes/units (bond k_b,r_0; angle k,theta_0; dihedral
 *     barrier_height/divider, phase, periodicity; LJ sigma,eps; mass):  src/ui/popup/ff_params.rs:322-580
 *   - per-term disable switches:                                        src/md/mod.rs:671-682
 *   - energy outputs and their units:                                   src/ui/panels/md_viewer.rs:195-257
 *   - velocity-Verlet integrator named at                               README.md:236-240
 * Conventions the absent crate fixes silently are explicit mdx_config fields (include/mdx.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/mdx.h"

#define ACC_CONV 418.4          /* kcal/mol/Å/Da -> Å/ps² */
#define KB_KCAL  0.0019872041   /* kcal/mol/K */

/* E_GLJ / E_GCOUL: the GROSS pair sums, sum over pairs of |e_pair| - the scale the fp32 rounding of the engine's pair terms acts on
 * (the net Coulomb energy of a neutral liquid is a difference of terms a thousand times larger); the parity tests' energy
 * tolerance for the pair sums is stated against them */
enum { E_BOND, E_ANGLE, E_DIHEDRAL, E_LJ, E_COUL, E_LJ14, E_COUL14, E_KIN, E_VIRIAL, E_CROSS, E_DUDL, E_GLJ, E_GCOUL, E_N };
/* E_CROSS: unscaled non-bonded energy between the alchemical molecule and the rest (0 without a window), at the soft-core
 * distance when the soft core is on.  E_DUDL: dU/dlambda of the window (= -E_CROSS for linear coupling). */
/* E_VIRIAL: W = sum_i r_i . F_i of the internal forces (pairs, 1-4, bonds; angle and dihedral terms are
 * scale invariant and contribute exactly 0), kcal/mol.  Pressure = (2 KE + W + W_constraints) / (3 V). */

int orc_num_energies(void) { return E_N; }
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int cutoff_on(float rc) { return rc > 0.0f && isfinite(rc); }

/* src/cuda/util.cu:65-71 / src/md/mod.rs:278-296 */
static inline void min_image(const mdx_system* s, double d[3]) {
    if (!s->periodic) return;
    for (int a = 0; a < 3; ++a) {
        double L = (double)s->box_hi[a] - (double)s->box_lo[a];
        if (L > 0.0) d[a] -= rint(d[a] / L) * L;
    }
}

/* Canonical fp32 pair distance: the same IEEE operations, in the same order, that the GPU's
 * neighbour-list extraction kernel performs (no FMA contraction: this file is built with
 * -ffp-contract=off and the fused steps are spelled fmaf). */
static inline float r2_canonical(const mdx_system* s, const float* pi, const float* pj) {
    float d[3];
    for (int a = 0; a < 3; ++a) {
        d[a] = pi[a] - pj[a];
        if (s->periodic) {
            float L = s->box_hi[a] - s->box_lo[a];
            d[a] = d[a] - rintf(d[a] / L) * L;
        }
    }
    return fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0]));
}

/* Test hooks: the canonical minimum image / distance on their own, so that tests can hold them against the reference's
 * `min_image` (src/cuda/util.cu:65-71, compiled into oracle/_ref) value by value, exact ties included. */
void orc_min_image_f32(const mdx_system* s, const float* d_in, float* d_out) {
    for (int a = 0; a < 3; ++a) {
        float d = d_in[a];
        if (s->periodic) {
            float L = s->box_hi[a] - s->box_lo[a];
            d = d - rintf(d / L) * L;
        }
        d_out[a] = d;
    }
}
float orc_r2_canonical(const mdx_system* s, const float* pi, const float* pj) { return r2_canonical(s, pi, pj); }

/* Position wrap into [lo, lo+L): identical fp32 sequence to the GPU binning kernel. */
void orc_wrap_f32(const mdx_system* s, float* pos, uint32_t n) {
    if (!s->periodic) return;
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            float lo = s->box_lo[a], L = s->box_hi[a] - s->box_lo[a];
            float x = pos[3 * i + a];
            float t = x - floorf((x - lo) / L) * L;
            if (t < lo) t += L;
            if (t >= lo + L) t -= L;
            pos[3 * i + a] = t;
        }
}

/* ------------------------------------------------------------------------------------------- */
/* exclusion lookup: merged (excl CSR + pairs14) per-atom sorted lists                          */
typedef struct { uint32_t* off; uint32_t* idx; } excl_t;

static int cmp_u32(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

static excl_t build_excl(const mdx_system* s) {
    uint32_t n = s->n_atoms;
    excl_t e;
    e.off = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i)
        e.off[i + 1] = s->excl_offsets ? s->excl_offsets[i + 1] - s->excl_offsets[i] : 0;
    for (uint32_t p = 0; p < s->n_pairs14; ++p) {
        e.off[s->pairs14_idx[2 * p] + 1]++;
        e.off[s->pairs14_idx[2 * p + 1] + 1]++;
    }
    for (uint32_t i = 0; i < n; ++i) e.off[i + 1] += e.off[i];
    e.idx = (uint32_t*)malloc(sizeof(uint32_t) * (e.off[n] ? e.off[n] : 1));
    uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    memcpy(cur, e.off, sizeof(uint32_t) * ((size_t)n + 1));
    if (s->excl_offsets)
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t k = s->excl_offsets[i]; k < s->excl_offsets[i + 1]; ++k)
                e.idx[cur[i]++] = s->excl_idx[k];
    for (uint32_t p = 0; p < s->n_pairs14; ++p) {
        uint32_t a = s->pairs14_idx[2 * p], b = s->pairs14_idx[2 * p + 1];
        e.idx[cur[a]++] = b;
        e.idx[cur[b]++] = a;
    }
    for (uint32_t i = 0; i < n; ++i)
        qsort(e.idx + e.off[i], e.off[i + 1] - e.off[i], sizeof(uint32_t), cmp_u32);
    free(cur);
    return e;
}

static inline int is_excluded(const excl_t* e, uint32_t i, uint32_t j) {
    uint32_t lo = e->off[i], hi = e->off[i + 1];
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (e->idx[mid] < j) lo = mid + 1; else hi = mid;
    }
    return lo < e->off[i + 1] && e->idx[lo] == j;
}

/* ------------------------------------------------------------------------------------------- */
/* pair parameters                                                                              */
static inline void lj_pair(const mdx_system* s, const mdx_config* c, uint32_t i, uint32_t j,
                           double* sig, double* eps) {
    double si = s->lj_sigma[s->lj_type[i]], sj = s->lj_sigma[s->lj_type[j]];
    double ei = s->lj_eps[s->lj_type[i]], ej = s->lj_eps[s->lj_type[j]];
    *sig = (c->combining_rule == MDX_COMBINE_GEOMETRIC) ? sqrt(si * sj) : 0.5 * (si + sj);
    *eps = sqrt(ei * ej);
}

static inline int nb_active(const mdx_system* s, uint32_t i) {
    return !(s->flags && (s->flags[i] & MDX_ATOM_BONDED_ONLY));
}

/* One non-bonded pair.  d = r_i - r_j (tgt - src, src/cuda/util.cu:118-140), r2 = |d|^2 (fp64).
 * Inclusion (r < rc) is decided by the caller from the canonical fp32 r2. Adds force on i
 * (caller adds the opposite on j) and energies. */
static inline void pair_terms(const mdx_config* c, double sig, double eps, double qq,
                              double r2, int in_lj, int in_coul, double* fs_out,
                              double* e_lj, double* e_coul) {
    double fs = 0.0; /* force on i = fs * d */
    double r = sqrt(r2), inv_r = 1.0 / r;
    if (in_lj && !(c->overrides & MDX_OVR_LJ_DISABLED) && eps != 0.0) {
        /* src/cuda/util.cu:92-115 */
        double sr = sig * inv_r, sr2 = sr * sr, sr6 = sr2 * sr2 * sr2, sr12 = sr6 * sr6;
        fs += 24.0 * eps * (2.0 * sr12 - sr6) * inv_r * inv_r;
        *e_lj += 4.0 * eps * (sr12 - sr6);
    }
    if (in_coul && !(c->overrides & MDX_OVR_COULOMB_DISABLED) && qq != 0.0) {
        double kqq = (double)c->coulomb_k * qq;
        double rc = (double)c->coulomb_cutoff;
        int cut = cutoff_on(c->coulomb_cutoff);
        switch (c->coulomb_mode) {
        case MDX_COULOMB_REACTION: {
            double krf = cut ? 1.0 / (2.0 * rc * rc * rc) : 0.0;
            double crf = cut ? 1.5 / rc : 0.0;
            fs += kqq * (inv_r * inv_r * inv_r - 2.0 * krf);
            *e_coul += kqq * (inv_r + krf * r2 - crf);
        } break;
        case MDX_COULOMB_EWALD: {
            double a = (double)c->ewald_alpha, ar = a * r;
            double erfc_ar = erfc(ar);
            fs += kqq * (erfc_ar * inv_r + 1.1283791670955126 * a * exp(-ar * ar)) * inv_r * inv_r;
            *e_coul += kqq * erfc_ar * inv_r;
        } break;
        default: /* MDX_COULOMB_SHIFTED; src/cuda/util.cu:53-63 with the softening as a parameter */
            fs += kqq * inv_r / (r2 + (double)c->softening_sq);
            *e_coul += kqq * (inv_r - (cut ? 1.0 / rc : 0.0));
        }
    }
    *fs_out = fs;
}

/* ------------------------------------------------------------------------------------------- */
/* bonded terms (Amber forms; ff_params.rs:352-372, 401-421, 474-511)                           */
static void bonded_forces(const mdx_system* s, const mdx_config* c, const double* x, double* f,
                          double* en) {
    if (c->overrides & MDX_OVR_BONDED_DISABLED) return;
    for (uint32_t b = 0; b < s->n_bonds; ++b) {
        uint32_t i = s->bond_idx[2 * b], j = s->bond_idx[2 * b + 1];
        double d[3] = { x[3*i] - x[3*j], x[3*i+1] - x[3*j+1], x[3*i+2] - x[3*j+2] };
        min_image(s, d);
        double r = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
        double dr = r - (double)s->bond_r0[b], k = s->bond_k[b];
        en[E_BOND] += k * dr * dr;
        double fs = -2.0 * k * dr / r;
        en[E_VIRIAL] += fs * r * r;
        for (int a = 0; a < 3; ++a) { f[3*i+a] += fs * d[a]; f[3*j+a] -= fs * d[a]; }
    }
    for (uint32_t t = 0; t < s->n_angles; ++t) {
        uint32_t i = s->angle_idx[3*t], j = s->angle_idx[3*t+1], k = s->angle_idx[3*t+2];
        double a1[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
        double a2[3] = { x[3*k]-x[3*j], x[3*k+1]-x[3*j+1], x[3*k+2]-x[3*j+2] };
        min_image(s, a1); min_image(s, a2);
        double r1 = sqrt(a1[0]*a1[0]+a1[1]*a1[1]+a1[2]*a1[2]);
        double r2 = sqrt(a2[0]*a2[0]+a2[1]*a2[1]+a2[2]*a2[2]);
        double cs = (a1[0]*a2[0]+a1[1]*a2[1]+a1[2]*a2[2]) / (r1 * r2);
        if (cs > 1.0) cs = 1.0; if (cs < -1.0) cs = -1.0;
        double th = acos(cs), dth = th - (double)s->angle_theta0[t], kk = s->angle_k[t];
        en[E_ANGLE] += kk * dth * dth;
        double sn = sqrt(1.0 - cs * cs); if (sn < 1e-8) sn = 1e-8;
        double dEdth = 2.0 * kk * dth;
        /* dtheta/dr_i = -(a2/|a2| - cos * a1/|a1|) / (|a1| sin) */
        for (int a = 0; a < 3; ++a) {
            double gi = -(a2[a] / r2 - cs * a1[a] / r1) / (r1 * sn);
            double gk = -(a1[a] / r1 - cs * a2[a] / r2) / (r2 * sn);
            f[3*i+a] -= dEdth * gi;
            f[3*k+a] -= dEdth * gk;
            f[3*j+a] += dEdth * (gi + gk);
        }
    }
    for (uint32_t t = 0; t < s->n_dihedrals; ++t) {
        uint32_t i = s->dihedral_idx[4*t], j = s->dihedral_idx[4*t+1],
                 k = s->dihedral_idx[4*t+2], l = s->dihedral_idx[4*t+3];
        /* Blondel & Karplus: F = ri-rj, G = rj-rk, H = rl-rk, A = F x G, B = H x G */
        double F[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
        double G[3] = { x[3*j]-x[3*k], x[3*j+1]-x[3*k+1], x[3*j+2]-x[3*k+2] };
        double H[3] = { x[3*l]-x[3*k], x[3*l+1]-x[3*k+1], x[3*l+2]-x[3*k+2] };
        min_image(s, F); min_image(s, G); min_image(s, H);
        double A[3] = { F[1]*G[2]-F[2]*G[1], F[2]*G[0]-F[0]*G[2], F[0]*G[1]-F[1]*G[0] };
        double B[3] = { H[1]*G[2]-H[2]*G[1], H[2]*G[0]-H[0]*G[2], H[0]*G[1]-H[1]*G[0] };
        double A2 = A[0]*A[0]+A[1]*A[1]+A[2]*A[2], B2 = B[0]*B[0]+B[1]*B[1]+B[2]*B[2];
        double Gn = sqrt(G[0]*G[0]+G[1]*G[1]+G[2]*G[2]);
        if (A2 < 1e-24 || B2 < 1e-24 || Gn < 1e-12) continue;
        double cosphi = (A[0]*B[0]+A[1]*B[1]+A[2]*B[2]);
        /* sin(phi) |A||B| = (B x A) . G / |G| */
        double BxA[3] = { B[1]*A[2]-B[2]*A[1], B[2]*A[0]-B[0]*A[2], B[0]*A[1]-B[1]*A[0] };
        double sinphi = (BxA[0]*G[0]+BxA[1]*G[1]+BxA[2]*G[2]) / Gn;
        double phi = atan2(sinphi, cosphi);
        double n = (double)s->dihedral_n[t], v = s->dihedral_v[t], ph = s->dihedral_phase[t];
        en[E_DIHEDRAL] += v * (1.0 + cos(n * phi - ph));
        double dEdphi = -v * n * sin(n * phi - ph);
        double FG = F[0]*G[0]+F[1]*G[1]+F[2]*G[2], HG = H[0]*G[0]+H[1]*G[1]+H[2]*G[2];
        for (int a = 0; a < 3; ++a) {
            double dpi = -Gn / A2 * A[a];
            double dpl =  Gn / B2 * B[a];
            double dpj =  Gn / A2 * A[a] + FG / (A2 * Gn) * A[a] - HG / (B2 * Gn) * B[a];
            double dpk = -Gn / B2 * B[a] - FG / (A2 * Gn) * A[a] + HG / (B2 * Gn) * B[a];
            f[3*i+a] -= dEdphi * dpi;
            f[3*j+a] -= dEdphi * dpj;
            f[3*k+a] -= dEdphi * dpk;
            f[3*l+a] -= dEdphi * dpl;
        }
    }
    /* scaled 1-4 pairs: no cutoff, no potential shift */
    for (uint32_t p = 0; p < s->n_pairs14; ++p) {
        uint32_t i = s->pairs14_idx[2*p], j = s->pairs14_idx[2*p+1];
        if (!nb_active(s, i) || !nb_active(s, j)) continue;
        double d[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
        min_image(s, d);
        double r2 = d[0]*d[0]+d[1]*d[1]+d[2]*d[2], r = sqrt(r2), inv_r = 1.0 / r;
        double sig, eps; lj_pair(s, c, i, j, &sig, &eps);
        double fs = 0.0;
        if (!(c->overrides & MDX_OVR_LJ_DISABLED)) {
            double sr = sig * inv_r, sr2 = sr*sr, sr6 = sr2*sr2*sr2, sr12 = sr6*sr6;
            fs += (double)c->scale14_lj * 24.0 * eps * (2.0 * sr12 - sr6) * inv_r * inv_r;
            en[E_LJ14] += (double)c->scale14_lj * 4.0 * eps * (sr12 - sr6);
        }
        if (!(c->overrides & MDX_OVR_COULOMB_DISABLED)) {
            double kqq = (double)c->scale14_coulomb * (double)c->coulomb_k *
                         (double)s->charge[i] * (double)s->charge[j];
            fs += kqq * inv_r * inv_r * inv_r;
            en[E_COUL14] += kqq * inv_r;
        }
        en[E_VIRIAL] += fs * r2;
        for (int a = 0; a < 3; ++a) { f[3*i+a] += fs * d[a]; f[3*j+a] -= fs * d[a]; }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* cell grid for O(N) pair search (periodic or not)                                             */
typedef struct {
    int n[3]; double lo[3], w[3]; uint32_t* start; uint32_t* items; int periodic;
} grid_t;

static grid_t build_grid(const mdx_system* s, const double* x, double rmin) {
    grid_t g; uint32_t N = s->n_atoms;
    g.periodic = s->periodic;
    double lo[3], hi[3];
    if (s->periodic) for (int a = 0; a < 3; ++a) { lo[a] = s->box_lo[a]; hi[a] = s->box_hi[a]; }
    else {
        for (int a = 0; a < 3; ++a) { lo[a] = 1e300; hi[a] = -1e300; }
        for (uint32_t i = 0; i < N; ++i) for (int a = 0; a < 3; ++a) {
            if (x[3*i+a] < lo[a]) lo[a] = x[3*i+a];
            if (x[3*i+a] > hi[a]) hi[a] = x[3*i+a];
        }
        for (int a = 0; a < 3; ++a) hi[a] += 1e-6;
    }
    size_t tot = 1;
    for (int a = 0; a < 3; ++a) {
        double L = hi[a] - lo[a];
        int n = (int)floor(L / rmin); if (n < 1) n = 1; if (n > 256) n = 256;
        g.n[a] = n; g.lo[a] = lo[a]; g.w[a] = L / n; tot *= (size_t)n;
    }
    g.start = (uint32_t*)calloc(tot + 1, sizeof(uint32_t));
    g.items = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    uint32_t* cell = (uint32_t*)malloc(sizeof(uint32_t) * (N ? N : 1));
    for (uint32_t i = 0; i < N; ++i) {
        int c[3];
        for (int a = 0; a < 3; ++a) {
            double L = g.w[a] * g.n[a], t = x[3*i+a] - g.lo[a];
            if (s->periodic) t -= floor(t / L) * L;
            int k = (int)floor(t / g.w[a]);
            if (k < 0) k = 0; if (k >= g.n[a]) k = g.n[a] - 1;
            c[a] = k;
        }
        cell[i] = (uint32_t)((c[2] * g.n[1] + c[1]) * g.n[0] + c[0]);
        g.start[cell[i] + 1]++;
    }
    for (size_t k = 0; k < tot; ++k) g.start[k + 1] += g.start[k];
    uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * (tot + 1));
    memcpy(cur, g.start, sizeof(uint32_t) * (tot + 1));
    for (uint32_t i = 0; i < N; ++i) g.items[cur[cell[i]]++] = i;
    free(cur); free(cell);
    return g;
}
static void free_grid(grid_t* g) { free(g->start); free(g->items); }

/* Enumerate the distinct neighbour cells of cell (cx,cy,cz) (27-stencil, deduplicated when a
 * dimension has < 3 cells). Returns count, fills ids. */
static int neighbour_cells(const grid_t* g, int cx, int cy, int cz, uint32_t* ids) {
    int cnt = 0;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        int c[3] = { cx + dx, cy + dy, cz + dz }, ok = 1;
        for (int a = 0; a < 3; ++a) {
            if (g->periodic) c[a] = ((c[a] % g->n[a]) + g->n[a]) % g->n[a];
            else if (c[a] < 0 || c[a] >= g->n[a]) ok = 0;
        }
        if (!ok) continue;
        uint32_t id = (uint32_t)((c[2] * g->n[1] + c[1]) * g->n[0] + c[0]);
        int dup = 0;
        for (int k = 0; k < cnt; ++k) if (ids[k] == id) { dup = 1; break; }
        if (!dup) ids[cnt++] = id;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------- */
/* Virtual sites r = r0 + a (r1 - r0) + b (r2 - r0) (OPC / TIP4P "M"; md.water[i].m,
 * src/properties/sol_shrinking_box.rs:605-613) and holonomic distance constraints
 * (HydrogenConstraint::Shake, src/ui/panels/md.rs:362-371): SHAKE positions, RATTLE velocities. */
void orc_vsite_construct(const mdx_system* s, double* x) {
    for (uint32_t i = 0; i < s->n_vsites; ++i) {
        uint32_t m = s->vsite_idx[4*i], p0 = s->vsite_idx[4*i+1], p1 = s->vsite_idx[4*i+2], p2 = s->vsite_idx[4*i+3];
        double a = s->vsite_w[2*i], b = s->vsite_w[2*i+1];
        double d1[3] = { x[3*p1]-x[3*p0], x[3*p1+1]-x[3*p0+1], x[3*p1+2]-x[3*p0+2] };
        double d2[3] = { x[3*p2]-x[3*p0], x[3*p2+1]-x[3*p0+1], x[3*p2+2]-x[3*p0+2] };
        min_image(s, d1); min_image(s, d2);
        for (int k = 0; k < 3; ++k) x[3*m+k] = x[3*p0+k] + a * d1[k] + b * d2[k];
    }
}
void orc_vsite_spread(const mdx_system* s, double* f) {
    for (uint32_t i = 0; i < s->n_vsites; ++i) {
        uint32_t m = s->vsite_idx[4*i], p0 = s->vsite_idx[4*i+1], p1 = s->vsite_idx[4*i+2], p2 = s->vsite_idx[4*i+3];
        double a = s->vsite_w[2*i], b = s->vsite_w[2*i+1];
        for (int k = 0; k < 3; ++k) {
            f[3*p0+k] += (1.0 - a - b) * f[3*m+k];
            f[3*p1+k] += a * f[3*m+k];
            f[3*p2+k] += b * f[3*m+k];
            f[3*m+k] = 0.0;
        }
    }
}
static double inv_mass(const mdx_system* s, uint32_t i) {
    if (s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST))) return 0.0;
    return 1.0 / (double)s->mass[i];
}
/* Constraint virial of the last SHAKE position stage (kcal/mol): a correction g*rv/m_a of atom a is the
 * work of a force G = 2 m_a dx_a / dt^2 = 2 g rv / dt^2 (it acts through the half kick and the drift of
 * velocity Verlet) along the old bond vector rv; sum r_i . G_i = 2 g |rv|^2 / dt^2 per correction. */
static double g_last_cons_virial = 0.0;
double orc_last_constraint_virial(void) { return g_last_cons_virial; }

/* SHAKE: x (new) corrected along the old bond vectors x_old; if v != NULL, v += dx/dt. */
int orc_constrain_positions(const mdx_system* s, double* x, const double* x_old, double* v, double dt, double tol) {
    double wc = 0.0;
    uint32_t N = s->n_atoms;
    double* x0 = (double*)malloc(sizeof(double) * 3 * N);
    memcpy(x0, x, sizeof(double) * 3 * N);
    int it;
    for (it = 0; it < 1000; ++it) {
        int done = 1;
        for (uint32_t c = 0; c < s->n_constraints; ++c) {
            uint32_t a = s->constraint_idx[2*c], b = s->constraint_idx[2*c+1];
            double l2 = (double)s->constraint_len[c] * s->constraint_len[c];
            double sv[3] = { x[3*a]-x[3*b], x[3*a+1]-x[3*b+1], x[3*a+2]-x[3*b+2] };
            double rv[3] = { x_old[3*a]-x_old[3*b], x_old[3*a+1]-x_old[3*b+1], x_old[3*a+2]-x_old[3*b+2] };
            min_image(s, sv); min_image(s, rv);
            double diff = l2 - (sv[0]*sv[0] + sv[1]*sv[1] + sv[2]*sv[2]);
            if (fabs(diff) > 2.0 * tol * l2) {
                done = 0;
                double ima = inv_mass(s, a), imb = inv_mass(s, b);
                double g = diff / (2.0 * (sv[0]*rv[0] + sv[1]*rv[1] + sv[2]*rv[2]) * (ima + imb));
                wc += g * (rv[0]*rv[0] + rv[1]*rv[1] + rv[2]*rv[2]);
                for (int k = 0; k < 3; ++k) { x[3*a+k] += g * ima * rv[k]; x[3*b+k] -= g * imb * rv[k]; }
            }
        }
        if (done) break;
    }
    g_last_cons_virial = dt != 0.0 ? 2.0 * wc / (dt * dt) / ACC_CONV : 0.0;
    if (v && dt != 0.0) for (uint32_t i = 0; i < 3 * N; ++i) v[i] += (x[i] - x0[i]) / dt;
    free(x0);
    return it;
}
/* RATTLE velocity stage: remove the relative velocity along every constrained bond. */
int orc_constrain_velocities(const mdx_system* s, const double* x, double* v, double tol) {
    int it;
    for (it = 0; it < 1000; ++it) {
        int done = 1;
        for (uint32_t c = 0; c < s->n_constraints; ++c) {
            uint32_t a = s->constraint_idx[2*c], b = s->constraint_idx[2*c+1];
            double l2 = (double)s->constraint_len[c] * s->constraint_len[c];
            double sv[3] = { x[3*a]-x[3*b], x[3*a+1]-x[3*b+1], x[3*a+2]-x[3*b+2] };
            min_image(s, sv);
            double dot = sv[0]*(v[3*a]-v[3*b]) + sv[1]*(v[3*a+1]-v[3*b+1]) + sv[2]*(v[3*a+2]-v[3*b+2]);
            if (fabs(dot) > tol * l2 * 10.0) {
                done = 0;
                double ima = inv_mass(s, a), imb = inv_mass(s, b);
                double g = dot / (l2 * (ima + imb));
                for (int k = 0; k < 3; ++k) { v[3*a+k] -= g * ima * sv[k]; v[3*b+k] += g * imb * sv[k]; }
            }
        }
        if (done) break;
    }
    return it;
}

/* Alchemical window (`md.configure_alchemical_window(dev, mol_index, lambda)`, src/properties/water_sol.rs:556):
 * linear coupling of the cut-off non-bonded pairs between atoms [lo, hi) and all others,
 * U(lambda) = U_rest + (1 - lambda) U_cross; en[E_CROSS] = U_cross, so dU/dlambda = -en[E_CROSS].
 * lambda < 0 switches it off.  Test infrastructure state, like the rest of this file. */
static double g_alch_lambda = -1.0; static uint32_t g_alch_lo = 0, g_alch_hi = 0;
void orc_set_alchemical(uint32_t lo, uint32_t hi, double lambda) { g_alch_lo = lo; g_alch_hi = hi; g_alch_lambda = lambda; }
/* Soft core (Beutler et al., Chem. Phys. Lett. 222, 529 (1994), the form GROMACS uses with sc-power 1, sc-r-power 6):
 * a cross pair interacts at r_sc = (alpha sigma^6 lambda + r^6)^(1/6) instead of r, LJ and Coulomb alike, and its energy
 * is scaled by (1 - lambda): U = U_rest + (1 - lambda) U_cross(r_sc(lambda)).  sigma = sigma_ij of the pair, or sigma_min
 * where the pair has no LJ interaction (sigma or eps zero: hydrogens).  Then
 *   dU/dlambda = -U_cross(r_sc) + (1 - lambda) U_cross'(r_sc) alpha sigma^6 / (6 r_sc^5),
 * finite at lambda = 1 whatever the overlap (r_sc >= (alpha sigma^6)^(1/6) there).  alpha = 0: linear coupling. */
static double g_sc_alpha = 0.0, g_sc_sigma_min = 3.0;
void orc_set_softcore(double alpha, double sigma_min) { g_sc_alpha = alpha; g_sc_sigma_min = sigma_min; }

/* ------------------------------------------------------------------------------------------- */
/* Forces + energies.  x: fp64 positions [3N] (NULL -> s->pos).  f: [3N] out.  en: [E_N] out.
 * use_cells: 0 = O(N^2) brute force, 1 = cell list (needs a cutoff).  Pair inclusion uses the
 * canonical fp32 r2 of the positions rounded to f32.  ext: optional external forces [3N]. */
int orc_forces(const mdx_system* s, const mdx_config* c, const double* x_in, const double* ext,
               double* f, double* en, int use_cells) {
    uint32_t N = s->n_atoms;
    double* x = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    float* xf = (float*)malloc(sizeof(float) * 3 * (N ? N : 1));
    for (uint32_t i = 0; i < 3 * N; ++i) {
        x[i] = x_in ? x_in[i] : (double)s->pos[i];
        xf[i] = (float)x[i];
    }
    orc_vsite_construct(s, x);
    for (uint32_t i = 0; i < 3 * N; ++i) xf[i] = (float)x[i];
    memset(f, 0, sizeof(double) * 3 * N);
    for (int k = 0; k < E_N; ++k) en[k] = 0.0;
    excl_t ex = build_excl(s);
    int cut_lj = cutoff_on(c->lj_cutoff), cut_c = cutoff_on(c->coulomb_cutoff);
    float rc2_lj = c->lj_cutoff * c->lj_cutoff, rc2_c = c->coulomb_cutoff * c->coulomb_cutoff;
    double rmax = fmax(cut_lj ? c->lj_cutoff : 0.0, cut_c ? c->coulomb_cutoff : 0.0);
    if (use_cells && !(cut_lj && cut_c)) use_cells = 0;

    grid_t g; if (use_cells) g = build_grid(s, x, rmax);
    double e_lj = 0.0, e_c = 0.0, w_nb = 0.0, e_x = 0.0, e_dl = 0.0, g_lj = 0.0, g_c = 0.0;
    const int alch = g_alch_lambda >= 0.0;
    const double asc = alch ? 1.0 - g_alch_lambda : 1.0;

#pragma omp parallel for schedule(dynamic, 64) reduction(+ : e_lj, e_c, w_nb, e_x, e_dl, g_lj, g_c)
    for (uint32_t i = 0; i < N; ++i) {
        if (!nb_active(s, i)) continue;
        double fi[3] = { 0, 0, 0 };
        /* full-list evaluation: each i sums over all j (every pair visited twice, energies halved);
         * keeps the loop race-free under OpenMP */
        uint32_t ids[27]; int ncell = 1; uint32_t jbeg = 0, jend = N;
        if (use_cells) {
            int cc[3];
            for (int a = 0; a < 3; ++a) {
                double L = g.w[a] * g.n[a], t = x[3*i+a] - g.lo[a];
                if (s->periodic) t -= floor(t / L) * L;
                int k = (int)floor(t / g.w[a]);
                if (k < 0) k = 0; if (k >= g.n[a]) k = g.n[a] - 1;
                cc[a] = k;
            }
            ncell = neighbour_cells(&g, cc[0], cc[1], cc[2], ids);
        }
        for (int ci = 0; ci < ncell; ++ci) {
            if (use_cells) { jbeg = g.start[ids[ci]]; jend = g.start[ids[ci] + 1]; }
            for (uint32_t jj = jbeg; jj < jend; ++jj) {
                uint32_t j = use_cells ? g.items[jj] : jj;
                if (j == i || !nb_active(s, j)) continue;
                float r2f = r2_canonical(s, xf + 3 * i, xf + 3 * j);
                int in_lj = !cut_lj || r2f < rc2_lj, in_c = !cut_c || r2f < rc2_c;
                if (!in_lj && !in_c) continue;
                if (is_excluded(&ex, i, j)) continue;
                double d[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
                min_image(s, d);
                double r2 = d[0]*d[0]+d[1]*d[1]+d[2]*d[2];
                double sig, eps; lj_pair(s, c, i, j, &sig, &eps);
                double qq = (double)s->charge[i] * (double)s->charge[j];
                double fs, el = 0.0, ec = 0.0;
                const int cross = alch && ((i >= g_alch_lo && i < g_alch_hi) != (j >= g_alch_lo && j < g_alch_hi));
                if (cross && g_sc_alpha > 0.0) {
                    const double sg = (sig > 0.0 && eps > 0.0) ? sig : g_sc_sigma_min;
                    const double sig6 = sg * sg * sg * sg * sg * sg;
                    const double rsc6 = g_sc_alpha * sig6 * g_alch_lambda + r2 * r2 * r2;
                    const double rsc2 = cbrt(rsc6), rsc4 = rsc2 * rsc2;
                    double fsc;
                    pair_terms(c, sig, eps, qq, rsc2, in_lj, in_c, &fsc, &el, &ec);   /* fsc = -U'(r_sc) / r_sc */
                    e_x += 0.5 * (el + ec);
                    e_dl += 0.5 * (-(el + ec) - asc * fsc * g_sc_alpha * sig6 / (6.0 * rsc4));
                    fs = asc * fsc * r2 * r2 / rsc4;                                  /* -dU/dr / r */
                    el *= asc; ec *= asc;
                } else {
                    pair_terms(c, sig, eps, qq, r2, in_lj, in_c, &fs, &el, &ec);
                    if (cross) {
                        e_x += 0.5 * (el + ec); e_dl -= 0.5 * (el + ec);
                        fs *= asc; el *= asc; ec *= asc;
                    }
                }
                fi[0] += fs * d[0]; fi[1] += fs * d[1]; fi[2] += fs * d[2];
                e_lj += 0.5 * el; e_c += 0.5 * ec; w_nb += 0.5 * fs * r2;
                g_lj += 0.5 * fabs(el); g_c += 0.5 * fabs(ec);
            }
        }
        f[3*i] = fi[0]; f[3*i+1] = fi[1]; f[3*i+2] = fi[2];
    }
    en[E_LJ] = e_lj; en[E_COUL] = e_c; en[E_VIRIAL] = w_nb; en[E_CROSS] = e_x; en[E_DUDL] = e_dl;
    en[E_GLJ] = g_lj; en[E_GCOUL] = g_c;
    bonded_forces(s, c, x, f, en);
    orc_vsite_spread(s, f);
    if (ext) for (uint32_t i = 0; i < 3 * N; ++i) f[i] += ext[i];
    if (use_cells) free_grid(&g);
    free(ex.off); free(ex.idx); free(x); free(xf);
    return 0;
}

/* SnapshotEnergyData.energy_potential_between_mols (consumed at /root/reference src/properties/crystal.rs:347-370, :533: flat
 * row-major n x n, upper triangle summed): the non-bonded energy between groups of atoms.  group[i] < G.  mat[a*G+b] = mat[b*G+a]
 * = sum over pairs (i in a, j in b) of the pair loop's LJ + Coulomb energy (same inclusion rule, images, exclusions and Coulomb
 * treatment as orc_forces) + the scaled 1-4 energy of 1-4 pairs between a and b; the diagonal holds the pairs inside a group,
 * each once.  gross: the same sums over |e_pair| (the scale fp32 rounding of the engine's pair terms acts on).  The alchemical
 * window, if set, scales the coupled pairs as orc_forces does. */
int orc_between_mols(const mdx_system* s, const mdx_config* c, const double* x_in, const uint8_t* group, uint32_t G,
                     double* mat, double* gross, int use_cells) {
    uint32_t N = s->n_atoms;
    double* x = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    float* xf = (float*)malloc(sizeof(float) * 3 * (N ? N : 1));
    for (uint32_t i = 0; i < 3 * N; ++i) x[i] = x_in ? x_in[i] : (double)s->pos[i];
    orc_vsite_construct(s, x);
    for (uint32_t i = 0; i < 3 * N; ++i) xf[i] = (float)x[i];
    for (uint32_t k = 0; k < G * G; ++k) { mat[k] = 0.0; gross[k] = 0.0; }
    excl_t ex = build_excl(s);
    int cut_lj = cutoff_on(c->lj_cutoff), cut_c = cutoff_on(c->coulomb_cutoff);
    float rc2_lj = c->lj_cutoff * c->lj_cutoff, rc2_c = c->coulomb_cutoff * c->coulomb_cutoff;
    double rmax = fmax(cut_lj ? c->lj_cutoff : 0.0, cut_c ? c->coulomb_cutoff : 0.0);
    if (use_cells && !(cut_lj && cut_c)) use_cells = 0;
    grid_t g; if (use_cells) g = build_grid(s, x, rmax);
    const int alch = g_alch_lambda >= 0.0;
    const double asc = alch ? 1.0 - g_alch_lambda : 1.0;
#pragma omp parallel
    {
        double* m_loc = (double*)calloc((size_t)G * G * 2, sizeof(double));
        double* g_loc = m_loc + (size_t)G * G;
#pragma omp for schedule(dynamic, 64)
        for (uint32_t i = 0; i < N; ++i) {
            if (!nb_active(s, i)) continue;
            uint32_t ids[27]; int ncell = 1; uint32_t jbeg = 0, jend = N;
            if (use_cells) {
                int cc[3];
                for (int a = 0; a < 3; ++a) {
                    double L = g.w[a] * g.n[a], t = x[3*i+a] - g.lo[a];
                    if (s->periodic) t -= floor(t / L) * L;
                    int k = (int)floor(t / g.w[a]);
                    if (k < 0) k = 0; if (k >= g.n[a]) k = g.n[a] - 1;
                    cc[a] = k;
                }
                ncell = neighbour_cells(&g, cc[0], cc[1], cc[2], ids);
            }
            for (int ci = 0; ci < ncell; ++ci) {
                if (use_cells) { jbeg = g.start[ids[ci]]; jend = g.start[ids[ci] + 1]; }
                for (uint32_t jj = jbeg; jj < jend; ++jj) {
                    uint32_t j = use_cells ? g.items[jj] : jj;
                    if (j <= i || !nb_active(s, j)) continue;          /* every pair once */
                    float r2f = r2_canonical(s, xf + 3 * i, xf + 3 * j);
                    int in_lj = !cut_lj || r2f < rc2_lj, in_c = !cut_c || r2f < rc2_c;
                    if (!in_lj && !in_c) continue;
                    if (is_excluded(&ex, i, j)) continue;
                    double d[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
                    min_image(s, d);
                    double r2 = d[0]*d[0]+d[1]*d[1]+d[2]*d[2];
                    double sig, eps; lj_pair(s, c, i, j, &sig, &eps);
                    double qq = (double)s->charge[i] * (double)s->charge[j];
                    double fs, el = 0.0, ec = 0.0;
                    const int cross = alch && ((i >= g_alch_lo && i < g_alch_hi) != (j >= g_alch_lo && j < g_alch_hi));
                    if (cross && g_sc_alpha > 0.0) {
                        const double sg = (sig > 0.0 && eps > 0.0) ? sig : g_sc_sigma_min;
                        const double sig6 = sg * sg * sg * sg * sg * sg;
                        const double rsc2 = cbrt(g_sc_alpha * sig6 * g_alch_lambda + r2 * r2 * r2);
                        pair_terms(c, sig, eps, qq, rsc2, in_lj, in_c, &fs, &el, &ec);
                    } else pair_terms(c, sig, eps, qq, r2, in_lj, in_c, &fs, &el, &ec);
                    if (cross) { el *= asc; ec *= asc; }
                    const uint32_t a = group[i], b = group[j];
                    const size_t k = a <= b ? (size_t)a * G + b : (size_t)b * G + a;      /* upper triangle, mirrored below */
                    m_loc[k] += el + ec; g_loc[k] += fabs(el) + fabs(ec);
                }
            }
        }
#pragma omp critical
        for (size_t k = 0; k < (size_t)G * G; ++k) { mat[k] += m_loc[k]; gross[k] += g_loc[k]; }
        free(m_loc);
    }
    /* scaled 1-4 pairs (bonded_forces above): no cutoff, no potential shift */
    if (!(c->overrides & MDX_OVR_BONDED_DISABLED))
        for (uint32_t p = 0; p < s->n_pairs14; ++p) {
            uint32_t i = s->pairs14_idx[2*p], j = s->pairs14_idx[2*p+1];
            if (!nb_active(s, i) || !nb_active(s, j)) continue;
            double d[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
            min_image(s, d);
            double r2 = d[0]*d[0]+d[1]*d[1]+d[2]*d[2], inv_r = 1.0 / sqrt(r2);
            double sig, eps; lj_pair(s, c, i, j, &sig, &eps);
            double e = 0.0, ga = 0.0;
            if (!(c->overrides & MDX_OVR_LJ_DISABLED)) {
                double sr = sig * inv_r, sr2 = sr*sr, sr6 = sr2*sr2*sr2, sr12 = sr6*sr6;
                double t = (double)c->scale14_lj * 4.0 * eps * (sr12 - sr6);
                e += t; ga += fabs(t);
            }
            if (!(c->overrides & MDX_OVR_COULOMB_DISABLED)) {
                double t = (double)c->scale14_coulomb * (double)c->coulomb_k * (double)s->charge[i] * (double)s->charge[j] * inv_r;
                e += t; ga += fabs(t);
            }
            const uint32_t a = group[i], b = group[j];
            const size_t k = a <= b ? (size_t)a * G + b : (size_t)b * G + a;
            mat[k] += e; gross[k] += ga;
        }
    for (uint32_t a = 0; a < G; ++a)
        for (uint32_t b = a + 1; b < G; ++b) { mat[(size_t)b * G + a] = mat[(size_t)a * G + b]; gross[(size_t)b * G + a] = gross[(size_t)a * G + b]; }
    if (use_cells) free_grid(&g);
    free(ex.off); free(ex.idx); free(x); free(xf);
    return 0;
}

/* Kinetic energy (kcal/mol) of fp64 velocities. */
double orc_kinetic(const mdx_system* s, const double* v) {
    double ke = 0.0;
    for (uint32_t i = 0; i < s->n_atoms; ++i) {
        if (s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST))) continue;
        ke += 0.5 * (double)s->mass[i] * (v[3*i]*v[3*i] + v[3*i+1]*v[3*i+1] + v[3*i+2]*v[3*i+2]);
    }
    return ke / ACC_CONV;
}

/* Velocity Verlet (README.md:236-240), fp64 state.  x, v updated in place.  en (E_N) receives the
 * energies of the final state.  Returns 0. */
int orc_step(const mdx_system* s, const mdx_config* c, double* x, double* v, double dt,
             uint32_t n_steps, const double* ext, double* en, int use_cells) {
    uint32_t N = s->n_atoms;
    double* f = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    double* xold = s->n_constraints ? (double*)malloc(sizeof(double) * 3 * (N ? N : 1)) : NULL;
    orc_forces(s, c, x, ext, f, en, use_cells);
    for (uint32_t st = 0; st < n_steps; ++st) {
        if (xold) memcpy(xold, x, sizeof(double) * 3 * N);
        for (uint32_t i = 0; i < N; ++i) {
            int fixed = s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
            double im = fixed ? 0.0 : ACC_CONV / (double)s->mass[i];
            for (int a = 0; a < 3; ++a) {
                v[3*i+a] += 0.5 * dt * f[3*i+a] * im;
                x[3*i+a] += fixed ? 0.0 : dt * v[3*i+a];
            }
        }
        if (xold) orc_constrain_positions(s, x, xold, v, dt, 1e-12);
        orc_vsite_construct(s, x);
        orc_forces(s, c, x, ext, f, en, use_cells);
        for (uint32_t i = 0; i < N; ++i) {
            int fixed = s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
            double im = fixed ? 0.0 : ACC_CONV / (double)s->mass[i];
            for (int a = 0; a < 3; ++a) v[3*i+a] += 0.5 * dt * f[3*i+a] * im;
        }
        if (xold) orc_constrain_velocities(s, x, v, 1e-12);
    }
    free(xold);
    en[E_KIN] = orc_kinetic(s, v);
    free(f);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Verlet neighbour list (caller atom order, rows ascending): all j != i with canonical fp32
 * r2 < rlist^2.  Two-call protocol like mdx_neighbor_list. pos: f32 [3N]. */
int orc_neighbor_list(const mdx_system* s, const float* pos, float rlist, uint32_t* offsets,
                      uint32_t* idx, int use_cells) {
    uint32_t N = s->n_atoms; float rl2 = rlist * rlist;
    grid_t g; double* xd = NULL;
    if (use_cells) {
        xd = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
        for (uint32_t i = 0; i < 3 * N; ++i) xd[i] = pos[i];
        g = build_grid(s, xd, (double)rlist * 1.0001);
    }
    uint32_t* cnt = (uint32_t*)calloc((size_t)N + 1, sizeof(uint32_t));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !idx) break;
#pragma omp parallel for schedule(dynamic, 64)
        for (uint32_t i = 0; i < N; ++i) {
            uint32_t ids[27]; int ncell = 1; uint32_t jbeg = 0, jend = N, k = 0;
            uint32_t* row = (pass == 1) ? idx + offsets[i] : NULL;
            if (use_cells) {
                int cc[3];
                for (int a = 0; a < 3; ++a) {
                    double L = g.w[a] * g.n[a], t = xd[3*i+a] - g.lo[a];
                    if (s->periodic) t -= floor(t / L) * L;
                    int q = (int)floor(t / g.w[a]);
                    if (q < 0) q = 0; if (q >= g.n[a]) q = g.n[a] - 1;
                    cc[a] = q;
                }
                ncell = neighbour_cells(&g, cc[0], cc[1], cc[2], ids);
            }
            for (int ci = 0; ci < ncell; ++ci) {
                if (use_cells) { jbeg = g.start[ids[ci]]; jend = g.start[ids[ci] + 1]; }
                for (uint32_t jj = jbeg; jj < jend; ++jj) {
                    uint32_t j = use_cells ? g.items[jj] : jj;
                    if (j == i) continue;
                    if (r2_canonical(s, pos + 3 * i, pos + 3 * j) < rl2) {
                        if (row) row[k] = j;
                        ++k;
                    }
                }
            }
            if (pass == 0) cnt[i] = k;
            else qsort(row, k, sizeof(uint32_t), cmp_u32);
        }
        if (pass == 0) {
            offsets[0] = 0;
            for (uint32_t i = 0; i < N; ++i) offsets[i + 1] = offsets[i] + cnt[i];
        }
    }
    free(cnt);
    if (use_cells) { free_grid(&g); free(xd); }
    return 0;
}

/* Atoms that have at least one non-excluded pair whose canonical r2 lies within a relative band
 * `rel` of a cutoff: a one-ulp difference in the distance arithmetic may flip such a pair in or
 * out.  slack[i] receives the summed magnitude of the pair forces at stake (kcal/mol/Å).  The GPU
 * parity tests widen the per-atom tolerance by exactly this amount. */
int orc_cutoff_slack(const mdx_system* s, const mdx_config* c, const float* pos, double rel,
                     double* slack) {
    uint32_t N = s->n_atoms;
    memset(slack, 0, sizeof(double) * N);
    int cut_lj = cutoff_on(c->lj_cutoff), cut_c = cutoff_on(c->coulomb_cutoff);
    if (!cut_lj && !cut_c) return 0;
    double rmax = fmax(cut_lj ? c->lj_cutoff : 0.0, cut_c ? c->coulomb_cutoff : 0.0);
    double* xd = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    for (uint32_t i = 0; i < 3 * N; ++i) xd[i] = pos[i];
    grid_t g = build_grid(s, xd, rmax * (1.0 + 2.0 * rel));
    excl_t ex = build_excl(s);
    double rl2 = (double)c->lj_cutoff * c->lj_cutoff, rcc2 = (double)c->coulomb_cutoff * c->coulomb_cutoff;
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t i = 0; i < N; ++i) {
        if (!nb_active(s, i)) continue;
        int cc[3]; uint32_t ids[27];
        for (int a = 0; a < 3; ++a) {
            double L = g.w[a] * g.n[a], t = xd[3*i+a] - g.lo[a];
            if (s->periodic) t -= floor(t / L) * L;
            int q = (int)floor(t / g.w[a]);
            if (q < 0) q = 0; if (q >= g.n[a]) q = g.n[a] - 1;
            cc[a] = q;
        }
        int ncell = neighbour_cells(&g, cc[0], cc[1], cc[2], ids);
        for (int ci = 0; ci < ncell; ++ci)
            for (uint32_t jj = g.start[ids[ci]]; jj < g.start[ids[ci] + 1]; ++jj) {
                uint32_t j = g.items[jj];
                if (j == i || !nb_active(s, j)) continue;
                double r2 = (double)r2_canonical(s, pos + 3 * i, pos + 3 * j);
                int near_lj = cut_lj && fabs(r2 - rl2) <= rel * rl2;
                int near_c = cut_c && fabs(r2 - rcc2) <= rel * rcc2;
                if (!near_lj && !near_c) continue;
                if (is_excluded(&ex, i, j)) continue;
                double sig, eps; lj_pair(s, c, i, j, &sig, &eps);
                double qq = (double)s->charge[i] * (double)s->charge[j];
                double fs, el = 0, ec = 0;
                pair_terms(c, sig, eps, qq, r2, near_lj, near_c, &fs, &el, &ec);
                slack[i] += fabs(fs) * sqrt(r2);
            }
    }
    free_grid(&g); free(ex.off); free(ex.idx); free(xd);
    return 0;
}

/* =============================================================================================
 * Callers either side of `step`: minimiser, velocity initialisation, thermostats.
 * The reference reaches these through MdState (src/ui/mol_editor.rs:375, src/mol_alignment.rs:356,
 * src/properties/sol_shrinking_box.rs:962-965, src/ui/panels/md.rs:296-305); their algorithms live
 * in the absent crate, so the ones restated here are the build's documented choice.
 * ============================================================================================= */
static uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rng_u01(uint64_t* s) { return ((double)(splitmix64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static double rng_gauss(uint64_t* s) {
    double u1 = rng_u01(s), u2 = rng_u01(s);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static double rng_gamma_int(uint64_t* s, long ia) {
    double d = (double)ia - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = rng_gauss(s), t = 1.0 + c * x;
        if (t <= 0.0) continue;
        double v = t * t * t, u = rng_u01(s);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v;
    }
}
static double rng_sum_noises(uint64_t* s, long nn) {
    if (nn <= 0) return 0.0;
    if (nn == 1) { double g = rng_gauss(s); return g * g; }
    if (nn % 2 == 0) return 2.0 * rng_gamma_int(s, nn / 2);
    double g = rng_gauss(s);
    return 2.0 * rng_gamma_int(s, (nn - 1) / 2) + g * g;
}

static uint32_t n_mobile(const mdx_system* s) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < s->n_atoms; ++i)
        if (!(s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST)))) ++n;
    return n;
}
double orc_dof(const mdx_system* s) { double d = 3.0 * n_mobile(s) - (double)s->n_constraints - 3.0; return d < 1.0 ? 1.0 : d; }

/* Maxwell-Boltzmann velocities: three normals per atom in caller order from splitmix64 + Box-Muller. */
void orc_init_velocities(const mdx_system* s, double temperature, int zero_com, uint64_t seed, double* v) {
    uint64_t st = seed;
    double p[3] = { 0, 0, 0 }, mt = 0;
    for (uint32_t i = 0; i < s->n_atoms; ++i) {
        int fixed = s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
        double sig = sqrt(KB_KCAL * temperature * ACC_CONV / (double)s->mass[i]);
        for (int a = 0; a < 3; ++a) {
            double g = rng_gauss(&st);
            v[3 * i + a] = fixed ? 0.0 : g * sig;
            if (!fixed) p[a] += (double)s->mass[i] * v[3 * i + a];
        }
        if (!fixed) mt += s->mass[i];
    }
    if (zero_com && mt > 0)
        for (uint32_t i = 0; i < s->n_atoms; ++i) {
            int fixed = s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
            if (!fixed) for (int a = 0; a < 3; ++a) v[3 * i + a] -= p[a] / mt;
        }
}

/* Velocity scaling factor of one thermostat application.  kind 1: Berendsen, 2: CSVR
 * (Bussi, Donadio, Parrinello 2007, eq. A7).  ke in kcal/mol, dt_couple in ps. */
double orc_thermostat_lambda(int kind, double ke, double nf, double temp_target, double tau, double dt_couple,
                             uint64_t* rng) {
    if (!(ke > 0.0)) return 1.0;
    double k0 = 0.5 * nf * KB_KCAL * temp_target;
    if (kind == 1) {
        double t = 2.0 * ke / (nf * KB_KCAL);
        double l2 = 1.0 + dt_couple / tau * (temp_target / t - 1.0);
        return sqrt(l2 > 0 ? l2 : 0);
    } else if (kind == 2) {
        double c = tau > 0 ? exp(-dt_couple / tau) : 0.0;
        double r1 = rng_gauss(rng);
        double sn = rng_sum_noises(rng, (long)nf - 1);
        double f = (1.0 - c) * k0 / (nf * ke);
        double a2 = c + f * (r1 * r1 + sn) + 2.0 * r1 * sqrt(c * f);
        return sqrt(a2 > 0 ? a2 : 0);
    }
    return 1.0;
}

/* Velocity Verlet with a thermostat applied every `every` steps (and optional COM drift removal at
 * the same cadence, before the rescale), mirroring mdx_step + mdx_after_steps. */
int orc_step_thermo(const mdx_system* s, const mdx_config* c, double* x, double* v, double dt, uint32_t n_steps,
                    int kind, double temp_target, double tau, uint32_t every, uint64_t seed, int zero_com,
                    double* temps_out /* [n_steps/every] or NULL */, int use_cells) {
    double en[E_N];
    uint64_t rng = seed;
    double nf = orc_dof(s);
    uint32_t k = 0;
    for (uint32_t st = 0; st < n_steps; st += every) {
        uint32_t n = every < n_steps - st ? every : n_steps - st;
        orc_step(s, c, x, v, dt, n, NULL, en, use_cells);
        if (n < every) break;
        if (zero_com) {
            double p[3] = { 0, 0, 0 }, mt = 0;
            for (uint32_t i = 0; i < s->n_atoms; ++i) {
                if (s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST))) continue;
                for (int a = 0; a < 3; ++a) p[a] += (double)s->mass[i] * v[3 * i + a];
                mt += s->mass[i];
            }
            for (uint32_t i = 0; i < s->n_atoms; ++i) {
                if (s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST))) continue;
                for (int a = 0; a < 3; ++a) v[3 * i + a] -= p[a] / mt;
            }
        }
        double ke = orc_kinetic(s, v);
        double lam = orc_thermostat_lambda(kind, ke, nf, temp_target, tau, dt * every, &rng);
        for (uint32_t i = 0; i < 3 * s->n_atoms; ++i) v[i] *= lam;
        if (temps_out) temps_out[k++] = 2.0 * orc_kinetic(s, v) / (nf * KB_KCAL);
    }
    return 0;
}

/* Three standard normals for (seed, step, atom): the stream mdx_integrate.hip draws from (24-bit uniforms
 * out of four splitmix64 words keyed by all three), Box-Muller evaluated in fp64 here. */
void orc_langevin_normals(uint64_t seed, uint64_t step, uint32_t atom, double* g) {
    uint64_t st = seed ^ (0x9E3779B97F4A7C15ull * (step + 1ull)) ^ (0xBF58476D1CE4E5B9ull * ((uint64_t)atom + 1ull));
    double u[4];
    for (int k = 0; k < 4; ++k) u[k] = ((double)(splitmix64(&st) >> 40) + 0.5) / 16777216.0;
    double r1 = sqrt(-2.0 * log(u[0])), r2 = sqrt(-2.0 * log(u[2]));
    g[0] = r1 * cos(6.283185307179586 * u[1]);
    g[1] = r1 * sin(6.283185307179586 * u[1]);
    g[2] = r2 * cos(6.283185307179586 * u[3]);
}

/* `Integrator::{Leapfrog (kind 1), LangevinMiddle{gamma} (kind 2)}` (src/ui/panels/md.rs:296-305) in fp64.
 * v holds half-step velocities.  Leapfrog: v += dt a; x += dt v.  Langevin middle (Zhang, Ding, Shang, Liu,
 * Leimkuhler 2019): v += dt a; x += dt/2 v; v = a1 v + sqrt(kT (1 - a1^2)/m) xi; x += dt/2 v, a1 = exp(-gamma dt).
 * Constraints: SHAKE along the bond vectors of x - dt v (exactly the previous positions for leapfrog), the
 * correction folded into v.  step0: global number of the first step (keys the noise).  en: final state. */
int orc_step_integrator(const mdx_system* s, const mdx_config* c, double* x, double* v, double dt, uint32_t n_steps,
                        int kind, double gamma, double temperature, uint64_t seed, uint64_t step0, double* en,
                        int use_cells) {
    uint32_t N = s->n_atoms;
    double* f = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    double* xo = s->n_constraints ? (double*)malloc(sizeof(double) * 3 * (N ? N : 1)) : NULL;
    double a1 = exp(-gamma * dt), ktn = KB_KCAL * temperature * (1.0 - a1 * a1);
    /* velocities handed in are projected onto the constraints first, as the engine does for any state it
     * receives (mdx ensure_ready) */
    if (xo) orc_constrain_velocities(s, x, v, 1e-12);
    orc_forces(s, c, x, NULL, f, en, use_cells);
    for (uint32_t st = 0; st < n_steps; ++st) {
        for (uint32_t i = 0; i < N; ++i) {
            int fixed = s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
            if (fixed) continue;
            double im = ACC_CONV / (double)s->mass[i];
            for (int a = 0; a < 3; ++a) v[3*i+a] += dt * f[3*i+a] * im;
            if (kind == 2) {
                double g[3]; orc_langevin_normals(seed, step0 + st, i, g);
                double sig = sqrt(ktn * im);
                for (int a = 0; a < 3; ++a) {
                    x[3*i+a] += 0.5 * dt * v[3*i+a];
                    v[3*i+a] = a1 * v[3*i+a] + sig * g[a];
                    x[3*i+a] += 0.5 * dt * v[3*i+a];
                }
            } else {
                for (int a = 0; a < 3; ++a) x[3*i+a] += dt * v[3*i+a];
            }
        }
        if (xo) {
            for (uint32_t i = 0; i < 3 * N; ++i) xo[i] = x[i] - dt * v[i];
            orc_constrain_positions(s, x, xo, v, dt, 1e-12);
            /* friction + noise sit between the half drifts: dx/dt is not an exact velocity projection */
            if (kind == 2) orc_constrain_velocities(s, x, v, 1e-12);
        }
        orc_vsite_construct(s, x);
        orc_forces(s, c, x, NULL, f, en, use_cells);
    }
    en[E_KIN] = orc_kinetic(s, v);
    free(f); free(xo);
    return 0;
}

/* Pressure (bar) of a state: en from orc_forces at x, ke in kcal/mol, w_cons from the last SHAKE. */
double orc_pressure(const mdx_system* s, const double* en, double ke, double w_cons) {
    if (!s->periodic) return 0.0;
    double V = ((double)s->box_hi[0] - s->box_lo[0]) * ((double)s->box_hi[1] - s->box_lo[1]) * ((double)s->box_hi[2] - s->box_lo[2]);
    return (2.0 * ke + en[E_VIRIAL] + w_cons) / (3.0 * V) * 69476.95;
}

/* Velocity Verlet with thermostat and a weak-coupling (Berendsen-style) barostat, both at their own
 * cadence, in the order mdx_after_steps applies them: COM removal, thermostat, barostat.  The
 * barostat scales the box edges and all coordinates about box_lo by
 * mu = cbrt(1 - beta (n dt / tau_p) (P0 - P)), |mu - 1| <= 1 %, then re-projects constrained clusters.
 * box_hi_io[3]: in = starting box_hi, out = final.  p_out/vol_out: one value per barostat application. */
int orc_step_npt(const mdx_system* s_in, const mdx_config* c, double* x, double* v, double dt, uint32_t n_steps,
                 int tkind, double temp_target, double tau_t, uint32_t every_t, uint64_t seed,
                 int bkind, double p0, double tau_p, double beta, uint32_t every_b,
                 float* box_hi_io, double* p_out, double* vol_out, int use_cells) {
    mdx_system s = *s_in;
    for (int a = 0; a < 3; ++a) s.box_hi[a] = box_hi_io[a];
    double en[E_N];
    uint64_t rng = seed;
    double nf = orc_dof(&s);
    uint32_t N = s.n_atoms, kb = 0;
    double* f = (double*)malloc(sizeof(double) * 3 * (N ? N : 1));
    if (!tkind) every_t = 0;
    if (!bkind) every_b = 0;
    uint32_t st = 0;
    while (st < n_steps) {
        uint32_t n = n_steps - st;
        if (every_t && every_t - st % every_t < n) n = every_t - st % every_t;
        if (every_b && every_b - st % every_b < n) n = every_b - st % every_b;
        orc_step(&s, c, x, v, dt, n, NULL, en, use_cells);
        st += n;
        if (every_t && st % every_t == 0) {
            double ke = orc_kinetic(&s, v);
            double lam = orc_thermostat_lambda(tkind, ke, nf, temp_target, tau_t, dt * every_t, &rng);
            for (uint32_t i = 0; i < 3 * N; ++i) v[i] *= lam;
        }
        if (every_b && st % every_b == 0) {
            double wc = s.n_constraints ? g_last_cons_virial : 0.0;
            orc_forces(&s, c, x, NULL, f, en, use_cells);
            double P = orc_pressure(&s, en, orc_kinetic(&s, v), wc);
            double mu3 = 1.0 - beta * dt * every_b / tau_p * (p0 - P);
            double mu = cbrt(mu3 > 0.5 ? mu3 : 0.5);
            if (mu > 1.01) mu = 1.01; if (mu < 0.99) mu = 0.99;
            for (uint32_t i = 0; i < N; ++i) for (int a = 0; a < 3; ++a)
                x[3*i+a] = (double)s.box_lo[a] + mu * (x[3*i+a] - (double)s.box_lo[a]);
            for (int a = 0; a < 3; ++a) s.box_hi[a] = s.box_lo[a] + (float)mu * (s.box_hi[a] - s.box_lo[a]);
            if (s.n_constraints) {
                double* xo = (double*)malloc(sizeof(double) * 3 * N);
                memcpy(xo, x, sizeof(double) * 3 * N);
                orc_constrain_positions(&s, x, xo, NULL, 0.0, 1e-12);
                orc_constrain_velocities(&s, x, v, 1e-12);
                free(xo);
            }
            orc_vsite_construct(&s, x);
            if (p_out) p_out[kb] = P;
            if (vol_out) vol_out[kb] = ((double)s.box_hi[0] - s.box_lo[0]) * ((double)s.box_hi[1] - s.box_lo[1]) * ((double)s.box_hi[2] - s.box_lo[2]);
            ++kb;
        }
    }
    for (int a = 0; a < 3; ++a) box_hi_io[a] = s.box_hi[a];
    free(f);
    return (int)kb;
}

/* Steepest descent with adaptive maximum displacement (same rule as mdx_minimize_energy).
 * x updated in place; returns the number of trial evaluations; e_out[E_N] = energies at the result. */
int orc_minimize(const mdx_system* s, const mdx_config* c, double* x, uint32_t max_iters, const double* ext,
                 double f_tol, double* e_out, int use_cells) {
    uint32_t N = s->n_atoms;
    double* f = (double*)malloc(sizeof(double) * 3 * N);
    double* ft = (double*)malloc(sizeof(double) * 3 * N);
    double* xt = (double*)malloc(sizeof(double) * 3 * N);
    double en[E_N], ent[E_N];
    orc_forces(s, c, x, ext, f, en, use_cells);
    double h = 0.01;
    uint32_t it = 0;
    for (;;) {
        double fmax = 0.0;
        for (uint32_t i = 0; i < N; ++i) {
            if (s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST))) continue;
            double m = sqrt(f[3*i]*f[3*i] + f[3*i+1]*f[3*i+1] + f[3*i+2]*f[3*i+2]);
            if (m > fmax) fmax = m;
        }
        if (it >= max_iters || fmax < f_tol || fmax <= 0.0) break;
        double scale = h / fmax;
        for (uint32_t i = 0; i < N; ++i) {
            int fixed = s->flags && (s->flags[i] & (MDX_ATOM_STATIC | MDX_ATOM_GHOST));
            for (int a = 0; a < 3; ++a) xt[3*i+a] = x[3*i+a] + (fixed ? 0.0 : scale * f[3*i+a]);
        }
        if (s->n_constraints) {   /* keep constrained bonds at their length (projection along the moved bonds) */
            double* tmp = (double*)malloc(sizeof(double) * 3 * N);
            memcpy(tmp, xt, sizeof(double) * 3 * N);
            orc_constrain_positions(s, xt, tmp, NULL, 0.0, 1e-12);
            free(tmp);
        }
        orc_vsite_construct(s, xt);
        orc_forces(s, c, xt, ext, ft, ent, use_cells);
        ++it;
        double ep = 0, ept = 0;
        for (int k = 0; k < E_KIN; ++k) { ep += en[k]; ept += ent[k]; }
        /* external forces (the alignment pull, src/mol_alignment.rs:356): what must drop is U - sum F_ext . x, i.e. the
           internal potential minus the work the external forces do along the trial move */
        if (ext) for (uint32_t i = 0; i < 3 * N; ++i) ept -= ext[i] * (xt[i] - x[i]);
        if (ept < ep) {
            memcpy(x, xt, sizeof(double) * 3 * N); memcpy(f, ft, sizeof(double) * 3 * N);
            memcpy(en, ent, sizeof(en));
            h *= 1.2; if (h > 0.2) h = 0.2;   /* ceiling on the largest displacement per iteration, as in mdx_minimize_energy */
        } else {
            h *= 0.5;
        }
    }
    memcpy(e_out, en, sizeof(en));
    free(f); free(ft); free(xt);
    return (int)it;
}
