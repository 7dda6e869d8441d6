/*
 * cpu_production.c — the CPU baseline bench.py times beside the GPU ("cpu_baseline.kind": "port-production").
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as mdx_oracle.c): nothing in the product path links or executes
 * this file; only tests/ and bench.py's cpu_baseline leg do.
 *
 * What it stands in for: the reference runs this path on `ComputationDevice::Cpu` through the external crate
 * `dynamics` (rayon + AVX, all cores — /root/reference README.md:208-211, src/util.rs:1072-1119), which is absent
 * and cannot be built here (no Rust toolchain).  SURVEY.md §8d therefore asks for this repo's own restatement in
 * fp32 "production mode": cell search, HALF Verlet list (each pair once, Newton's third law), the list REUSED across
 * steps until an atom has moved skin/2, OpenMP over all host cores, -O3 -march=native.  The fp64 oracle
 * (mdx_oracle.c) stays the correctness checker; tests/test_cpu_production.py pins this file's forces, energies and
 * trajectory against it.  Formulas and conventions are those of mdx_oracle.c (LJ 12-6 and the tgt - src direction of
 * src/cuda/util.cu:92-140, Coulomb form :53-63, minimum image by rint :65-71, Amber bonded forms).
 *
 * Supported: orthorhombic periodic systems, MDX_COULOMB_SHIFTED / MDX_COULOMB_REACTION, separate LJ / Coulomb
 * cut-offs, exclusions and scaled 1-4 pairs, bonds / angles / dihedrals, static atoms, velocity Verlet.
 * Refused (NULL / -1): vacuum, Ewald, distance constraints, virtual sites.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/mdx.h"

#define ACC_CONV 418.4f
#define BLK 2048u            /* granularity (atoms) at which a thread's private force buffer is tracked */

enum { PE_BOND, PE_ANGLE, PE_DIHEDRAL, PE_LJ, PE_COUL, PE_LJ14, PE_COUL14, PE_KIN, PE_N };

typedef struct {
    const mdx_system* s; const mdx_config* c;
    uint32_t N; int T;
    float L[3], invL[3], lo[3];
    float rc2_lj, rc2_coul, rl2, half_skin2;
    float *c12, *c6;                 /* [T*T] 4 eps sig^12, 4 eps sig^6 */
    float *qs;                       /* q * sqrt(ke) */
    float *invm;                     /* ACC_CONV / m, 0 = static */
    /* exclusions (merged with 1-4), sorted per atom */
    uint32_t *ex_off, *ex_idx;
    /* half Verlet list, CSR over i, j > i */
    uint64_t *nl_off; uint32_t *nl_idx; uint64_t nl_cap;
    float *xref;
    /* per-thread force buffers, tracked in blocks */
    int nthreads; float **fb; uint8_t **touched; uint32_t nblk;
    /* static partition of the pair rows: thread t owns rows [row_lo[t], row_lo[t+1]) - equal PAIR counts, contiguous in the
     * (spatially coherent) atom order, so the j atoms it writes to lie in one window of blocks [blk_lo[t], blk_hi[t]) */
    uint32_t *row_lo; uint32_t *blk_lo, *blk_hi;
    uint64_t pairs_evaluated;
    uint32_t rebuilds;
} prod_t;

static int thread_count(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
int cpu_prod_max_threads(void) { return thread_count(); }
static uint64_t g_last_pairs = 0;   /* list pairs evaluated by the last cpu_prod_run (all force passes) */
uint64_t cpu_prod_last_pairs(void) { return g_last_pairs; }

static int cmp_u32(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

static void build_exclusions(prod_t* p) {
    const mdx_system* s = p->s; uint32_t n = p->N;
    p->ex_off = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) p->ex_off[i + 1] = s->excl_offsets ? s->excl_offsets[i + 1] - s->excl_offsets[i] : 0;
    for (uint32_t k = 0; k < s->n_pairs14; ++k) { p->ex_off[s->pairs14_idx[2 * k] + 1]++; p->ex_off[s->pairs14_idx[2 * k + 1] + 1]++; }
    for (uint32_t i = 0; i < n; ++i) p->ex_off[i + 1] += p->ex_off[i];
    p->ex_idx = (uint32_t*)malloc(sizeof(uint32_t) * (p->ex_off[n] ? p->ex_off[n] : 1));
    uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    memcpy(cur, p->ex_off, sizeof(uint32_t) * ((size_t)n + 1));
    if (s->excl_offsets)
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t k = s->excl_offsets[i]; k < s->excl_offsets[i + 1]; ++k) p->ex_idx[cur[i]++] = s->excl_idx[k];
    for (uint32_t k = 0; k < s->n_pairs14; ++k) {
        uint32_t a = s->pairs14_idx[2 * k], b = s->pairs14_idx[2 * k + 1];
        p->ex_idx[cur[a]++] = b; p->ex_idx[cur[b]++] = a;
    }
    for (uint32_t i = 0; i < n; ++i) qsort(p->ex_idx + p->ex_off[i], p->ex_off[i + 1] - p->ex_off[i], sizeof(uint32_t), cmp_u32);
    free(cur);
}

static inline int excluded(const prod_t* p, uint32_t i, uint32_t j) {
    for (uint32_t k = p->ex_off[i]; k < p->ex_off[i + 1]; ++k) if (p->ex_idx[k] == j) return 1;   /* rows are a handful long */
    return 0;
}

static prod_t* prod_create(const mdx_system* s, const mdx_config* c) {
    if (!s->periodic || c->coulomb_mode == MDX_COULOMB_EWALD) return NULL;
    /* rigid molecules and virtual sites are not restated here: timing such a system without them would time a different
     * system (round-2 advisor finding) */
    if (s->n_constraints || s->n_vsites) return NULL;
    prod_t* p = (prod_t*)calloc(1, sizeof(prod_t));
    p->s = s; p->c = c; p->N = s->n_atoms; p->T = (int)s->n_lj_types;
    for (int a = 0; a < 3; ++a) { p->L[a] = s->box_hi[a] - s->box_lo[a]; p->invL[a] = 1.0f / p->L[a]; p->lo[a] = s->box_lo[a]; }
    p->rc2_lj = c->lj_cutoff * c->lj_cutoff; p->rc2_coul = c->coulomb_cutoff * c->coulomb_cutoff;
    const float rl = fmaxf(c->lj_cutoff, c->coulomb_cutoff) + c->skin;
    p->rl2 = rl * rl; p->half_skin2 = 0.25f * c->skin * c->skin;
    const int T = p->T;
    p->c12 = (float*)malloc(sizeof(float) * T * T); p->c6 = (float*)malloc(sizeof(float) * T * T);
    const int lj_off = (c->overrides & MDX_OVR_LJ_DISABLED) != 0, coul_off = (c->overrides & MDX_OVR_COULOMB_DISABLED) != 0;
    for (int a = 0; a < T; ++a)
        for (int b = 0; b < T; ++b) {
            double sg = c->combining_rule == MDX_COMBINE_GEOMETRIC ? sqrt((double)s->lj_sigma[a] * s->lj_sigma[b])
                                                                   : 0.5 * ((double)s->lj_sigma[a] + s->lj_sigma[b]);
            double ep = lj_off ? 0.0 : sqrt((double)s->lj_eps[a] * s->lj_eps[b]);
            double s6 = sg * sg * sg * sg * sg * sg;
            p->c12[a * T + b] = (float)(4.0 * ep * s6 * s6); p->c6[a * T + b] = (float)(4.0 * ep * s6);
        }
    p->qs = (float*)malloc(sizeof(float) * p->N); p->invm = (float*)malloc(sizeof(float) * p->N);
    const float sk = sqrtf(c->coulomb_k);
    for (uint32_t i = 0; i < p->N; ++i) {
        const uint8_t fl = s->flags ? s->flags[i] : 0;
        p->qs[i] = (coul_off || (fl & MDX_ATOM_BONDED_ONLY)) ? 0.f : s->charge[i] * sk;
        p->invm[i] = (fl & (MDX_ATOM_STATIC | MDX_ATOM_GHOST)) ? 0.f : ACC_CONV / s->mass[i];
    }
    build_exclusions(p);
    p->nl_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)p->N + 1));
    p->xref = (float*)malloc(sizeof(float) * 3 * (size_t)p->N);
    p->nthreads = thread_count();
    p->nblk = (p->N + BLK - 1) / BLK;
    p->fb = (float**)calloc(p->nthreads, sizeof(float*)); p->touched = (uint8_t**)calloc(p->nthreads, sizeof(uint8_t*));
    for (int t = 0; t < p->nthreads; ++t) {
        p->fb[t] = (float*)calloc((size_t)p->nblk * BLK * 3, sizeof(float));   /* pages are committed when first touched */
        p->touched[t] = (uint8_t*)calloc(p->nblk, 1);
    }
    p->row_lo = (uint32_t*)calloc((size_t)p->nthreads + 1, sizeof(uint32_t));
    p->blk_lo = (uint32_t*)calloc((size_t)p->nthreads, sizeof(uint32_t));
    p->blk_hi = (uint32_t*)calloc((size_t)p->nthreads, sizeof(uint32_t));
    return p;
}

static void prod_destroy(prod_t* p) {
    if (!p) return;
    for (int t = 0; t < p->nthreads; ++t) { free(p->fb[t]); free(p->touched[t]); }
    free(p->fb); free(p->touched); free(p->c12); free(p->c6); free(p->qs); free(p->invm);
    free(p->ex_off); free(p->ex_idx); free(p->nl_off); free(p->nl_idx); free(p->xref); free(p->row_lo); free(p->blk_lo); free(p->blk_hi); free(p);
}

/* ---- cell search -> half Verlet list ------------------------------------------------------------------------ */
static void prod_rebuild(prod_t* p, const float* x) {
    const uint32_t N = p->N;
    const float rl = sqrtf(p->rl2);
    int nc[3]; float w[3];
    for (int a = 0; a < 3; ++a) {          /* cells of edge >= r_list / 2: a 5^3 stencil covers the list radius */
        nc[a] = (int)floorf(p->L[a] / (0.5f * rl)); if (nc[a] < 5) nc[a] = 5; if (nc[a] > 512) nc[a] = 512;
        w[a] = p->L[a] / nc[a];
    }
    const int reach[3] = { (int)ceilf(rl / w[0]), (int)ceilf(rl / w[1]), (int)ceilf(rl / w[2]) };
    const size_t ncell = (size_t)nc[0] * nc[1] * nc[2];
    uint32_t* start = (uint32_t*)calloc(ncell + 1, sizeof(uint32_t));
    uint32_t* cell = (uint32_t*)malloc(sizeof(uint32_t) * N);
    uint32_t* items = (uint32_t*)malloc(sizeof(uint32_t) * N);
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < N; ++i) {
        int cc[3];
        for (int a = 0; a < 3; ++a) {
            float t = x[3 * i + a] - p->lo[a];
            t -= floorf(t * p->invL[a]) * p->L[a];
            int k = (int)(t / w[a]); if (k < 0) k = 0; if (k >= nc[a]) k = nc[a] - 1;
            cc[a] = k;
        }
        cell[i] = (uint32_t)((cc[2] * nc[1] + cc[1]) * nc[0] + cc[0]);
    }
    for (uint32_t i = 0; i < N; ++i) start[cell[i] + 1]++;
    for (size_t k = 0; k < ncell; ++k) start[k + 1] += start[k];
    {
        uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * (ncell + 1));
        memcpy(cur, start, sizeof(uint32_t) * (ncell + 1));
        for (uint32_t i = 0; i < N; ++i) items[cur[cell[i]]++] = i;
        free(cur);
    }
    /* cell-sorted SoA copy of the coordinates: the search streams contiguous runs (vectorisable) */
    float* sx = (float*)malloc(sizeof(float) * N); float* sy = (float*)malloc(sizeof(float) * N); float* sz = (float*)malloc(sizeof(float) * N);
#pragma omp parallel for schedule(static)
    for (uint32_t k = 0; k < N; ++k) { const uint32_t i = items[k]; sx[k] = x[3 * i]; sy[k] = x[3 * i + 1]; sz[k] = x[3 * i + 2]; }
    const float Lx = p->L[0], Ly = p->L[1], Lz = p->L[2], ix = p->invL[0], iy = p->invL[1], iz = p->invL[2], rl2 = p->rl2;
    /* two passes over the same search: count, then fill */
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            uint64_t tot = 0;
            for (uint32_t i = 0; i < N; ++i) { uint64_t n = p->nl_off[i]; p->nl_off[i] = tot; tot += n; }
            p->nl_off[N] = tot;
            if (tot > p->nl_cap) { free(p->nl_idx); p->nl_cap = tot + tot / 8 + 1024; p->nl_idx = (uint32_t*)malloc(sizeof(uint32_t) * p->nl_cap); }
        }
#pragma omp parallel for schedule(dynamic, 256)
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t ci = cell[i];
            const int cx = (int)(ci % nc[0]), cy = (int)((ci / nc[0]) % nc[1]), cz = (int)(ci / ((uint32_t)nc[0] * nc[1]));
            const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
            const int has_excl = p->ex_off[i + 1] != p->ex_off[i];
            uint64_t n = 0; uint32_t* out = pass ? p->nl_idx + p->nl_off[i] : NULL;
            for (int dz = -reach[2]; dz <= reach[2]; ++dz)
                for (int dy = -reach[1]; dy <= reach[1]; ++dy) {
                    if (2 * reach[1] + 1 > nc[1] && (dy < -(nc[1] / 2) || dy > (nc[1] - 1) / 2)) continue;   /* stencil wraps onto itself */
                    if (2 * reach[2] + 1 > nc[2] && (dz < -(nc[2] / 2) || dz > (nc[2] - 1) / 2)) continue;
                    const int ay = (cy + dy + nc[1]) % nc[1], az = (cz + dz + nc[2]) % nc[2];
                    for (int dx = -reach[0]; dx <= reach[0]; ++dx) {
                        if (2 * reach[0] + 1 > nc[0] && (dx < -(nc[0] / 2) || dx > (nc[0] - 1) / 2)) continue;
                        const int ax = (cx + dx + nc[0]) % nc[0];
                        const size_t cj = ((size_t)az * nc[1] + ay) * nc[0] + ax;
                        const uint32_t k0 = start[cj], k1 = start[cj + 1];
                        if (!pass) {
                            uint32_t m = 0;
#pragma omp simd reduction(+ : m)
                            for (uint32_t k = k0; k < k1; ++k) {
                                float ddx = xi - sx[k], ddy = yi - sy[k], ddz = zi - sz[k];
                                ddx -= rintf(ddx * ix) * Lx; ddy -= rintf(ddy * iy) * Ly; ddz -= rintf(ddz * iz) * Lz;
                                m += (ddx * ddx + ddy * ddy + ddz * ddz < rl2) & (items[k] > i);
                            }
                            n += m;
                        } else {
                            for (uint32_t k = k0; k < k1; ++k) {
                                float ddx = xi - sx[k], ddy = yi - sy[k], ddz = zi - sz[k];
                                ddx -= rintf(ddx * ix) * Lx; ddy -= rintf(ddy * iy) * Ly; ddz -= rintf(ddz * iz) * Lz;
                                const uint32_t j = items[k];
                                if (!(ddx * ddx + ddy * ddy + ddz * ddz < rl2) || j <= i) continue;
                                if (has_excl && excluded(p, i, j)) continue;
                                out[n++] = j;
                            }
                        }
                    }
                }
            if (!pass) {
                if (has_excl)      /* the count above included this atom's excluded partners inside the list radius */
                    for (uint32_t k = p->ex_off[i]; k < p->ex_off[i + 1]; ++k) {
                        const uint32_t j = p->ex_idx[k];
                        if (j <= i) continue;
                        float ddx = xi - x[3 * j], ddy = yi - x[3 * j + 1], ddz = zi - x[3 * j + 2];
                        ddx -= rintf(ddx * ix) * Lx; ddy -= rintf(ddy * iy) * Ly; ddz -= rintf(ddz * iz) * Lz;
                        if (ddx * ddx + ddy * ddy + ddz * ddz < rl2) --n;
                    }
                p->nl_off[i] = n;
            }
        }
    }
    free(sx); free(sy); free(sz);
    {   /* rows -> threads by equal pair counts; the window of blocks each thread's rows write to (j > i: it starts at its first row) */
        const uint64_t tot = p->nl_off[N];
        uint32_t r = 0;
        for (int t = 0; t < p->nthreads; ++t) {
            p->row_lo[t] = r;
            const uint64_t goal = tot * (uint64_t)(t + 1) / (uint64_t)p->nthreads;
            while (r < N && p->nl_off[r + 1] <= goal) ++r;
            if (t == p->nthreads - 1) r = N;
        }
        p->row_lo[p->nthreads] = N;
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < p->nthreads; ++t) {
            const uint32_t a = p->row_lo[t], b = p->row_lo[t + 1];
            uint32_t mx = b ? b - 1 : 0;
            for (uint64_t k = p->nl_off[a]; k < p->nl_off[b]; ++k) if (p->nl_idx[k] > mx) mx = p->nl_idx[k];
            p->blk_lo[t] = a / BLK; p->blk_hi[t] = (a < b) ? mx / BLK + 1 : a / BLK;
        }
    }
    memcpy(p->xref, x, sizeof(float) * 3 * (size_t)N);
    free(start); free(cell); free(items);
    p->rebuilds++;
}

/* ---- forces ---------------------------------------------------------------------------------------------------- */
static inline void touch(uint8_t* t, uint32_t i) { t[i / BLK] = 1; }

static inline __attribute__((always_inline)) void pair_rows(prod_t* p, const float* x, const uint32_t* type, int want_e,
                                                              double* en) {
    const uint32_t N = p->N; const int T = p->T;
    const mdx_config* c = p->c;
    const float Lx = p->L[0], Ly = p->L[1], Lz = p->L[2], ix = p->invL[0], iy = p->invL[1], iz = p->invL[2];
    const float rc2l = p->rc2_lj, rc2c = p->rc2_coul, soft = c->softening_sq;
    const int rf = c->coulomb_mode == MDX_COULOMB_REACTION;
    const float rc = c->coulomb_cutoff;
    const float krf2 = rf ? 1.0f / (rc * rc * rc) : 0.f, krf = 0.5f * krf2, crf = rf ? 1.5f / rc : 1.0f / rc;
    double e_lj = 0.0, e_c = 0.0;
#pragma omp parallel reduction(+ : e_lj, e_c)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        float* fb = p->fb[tid]; uint8_t* tb = p->touched[tid];
        /* (the team may be smaller than the partition was built for: a thread then takes every nthreads-th share) */
#ifdef _OPENMP
        const int team = omp_get_num_threads();
#else
        const int team = 1;
#endif
        for (int share = tid; share < p->nthreads; share += team) {
        for (uint32_t bb = p->blk_lo[share]; bb < p->blk_hi[share]; ++bb) tb[bb] = 1;
        for (uint32_t i = p->row_lo[share]; i < p->row_lo[share + 1]; ++i) {
            const uint64_t a = p->nl_off[i], b = p->nl_off[i + 1];
            if (a == b) continue;
            const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2], qi = p->qs[i];
            const float* c12r = p->c12 + (size_t)type[i] * T; const float* c6r = p->c6 + (size_t)type[i] * T;
            float fx = 0.f, fy = 0.f, fz = 0.f, elj = 0.f, ec = 0.f;
            const uint32_t* nl = p->nl_idx + a; const uint32_t cnt = (uint32_t)(b - a);
#pragma omp simd reduction(+ : fx, fy, fz, elj, ec)
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t j = nl[k];
                float dx = xi - x[3 * j], dy = yi - x[3 * j + 1], dz = zi - x[3 * j + 2];
                dx -= rintf(dx * ix) * Lx; dy -= rintf(dy * iy) * Ly; dz -= rintf(dz * iz) * Lz;
                const float r2 = dx * dx + dy * dy + dz * dz;
                const float inv_r = 1.0f / sqrtf(r2), inv_r2 = inv_r * inv_r;
                const float ml = r2 < rc2l ? 1.f : 0.f, mc = r2 < rc2c ? 1.f : 0.f;
                const float r6 = inv_r2 * inv_r2 * inv_r2;
                const float c12 = c12r[type[j]] * ml, c6 = c6r[type[j]] * ml;
                float fs = (12.f * c12 * r6 - 6.f * c6) * r6 * inv_r2;
                const float qq = qi * p->qs[j] * mc;
                fs += rf ? qq * (inv_r * inv_r2 - krf2) : qq * inv_r / (r2 + soft);
                if (want_e) {
                    elj += (c12 * r6 - c6) * r6;
                    ec += rf ? qq * (inv_r + krf * r2 - crf) : qq * (inv_r - crf);
                }
                const float gx = fs * dx, gy = fs * dy, gz = fs * dz;
                fx += gx; fy += gy; fz += gz;
                fb[3 * j] -= gx; fb[3 * j + 1] -= gy; fb[3 * j + 2] -= gz;     /* j is unique within a row */
            }
            fb[3 * i] += fx; fb[3 * i + 1] += fy; fb[3 * i + 2] += fz;
            e_lj += elj; e_c += ec;
        }
        }
    }
    p->pairs_evaluated += p->nl_off[N];
    if (want_e) { en[PE_LJ] += e_lj; en[PE_COUL] += e_c; }
}

static inline void mimg(const prod_t* p, float d[3]) {
    for (int a = 0; a < 3; ++a) d[a] -= rintf(d[a] * p->invL[a]) * p->L[a];
}

static void bonded(prod_t* p, const float* x, int want_e, double* en) {
    const mdx_system* s = p->s; const mdx_config* c = p->c;
    if (c->overrides & MDX_OVR_BONDED_DISABLED) return;
    double eb = 0, ea = 0, ed = 0, e14l = 0, e14c = 0;
    const int T = p->T;
#pragma omp parallel reduction(+ : eb, ea, ed, e14l, e14c)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        float* f = p->fb[tid]; uint8_t* tb = p->touched[tid];
#pragma omp for schedule(static) nowait
        for (uint32_t b = 0; b < s->n_bonds; ++b) {
            const uint32_t i = s->bond_idx[2 * b], j = s->bond_idx[2 * b + 1];
            float d[3] = { x[3*i] - x[3*j], x[3*i+1] - x[3*j+1], x[3*i+2] - x[3*j+2] };
            mimg(p, d);
            const float r = sqrtf(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]), dr = r - s->bond_r0[b], k = s->bond_k[b];
            eb += (double)(k * dr * dr);
            const float fs = -2.0f * k * dr / r;
            for (int a = 0; a < 3; ++a) { f[3*i+a] += fs * d[a]; f[3*j+a] -= fs * d[a]; }
            touch(tb, i); touch(tb, j);
        }
#pragma omp for schedule(static) nowait
        for (uint32_t t = 0; t < s->n_angles; ++t) {
            const uint32_t i = s->angle_idx[3*t], j = s->angle_idx[3*t+1], k = s->angle_idx[3*t+2];
            float a1[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
            float a2[3] = { x[3*k]-x[3*j], x[3*k+1]-x[3*j+1], x[3*k+2]-x[3*j+2] };
            mimg(p, a1); mimg(p, a2);
            const float r1 = sqrtf(a1[0]*a1[0]+a1[1]*a1[1]+a1[2]*a1[2]), r2 = sqrtf(a2[0]*a2[0]+a2[1]*a2[1]+a2[2]*a2[2]);
            float cs = (a1[0]*a2[0]+a1[1]*a2[1]+a1[2]*a2[2]) / (r1 * r2);
            cs = fminf(1.f, fmaxf(-1.f, cs));
            const float th = acosf(cs), dth = th - s->angle_theta0[t], kk = s->angle_k[t];
            ea += (double)(kk * dth * dth);
            const float sn = fmaxf(sqrtf(1.f - cs * cs), 1e-6f), de = 2.f * kk * dth;
            for (int a = 0; a < 3; ++a) {
                const float gi = -(a2[a] / r2 - cs * a1[a] / r1) / (r1 * sn), gk = -(a1[a] / r1 - cs * a2[a] / r2) / (r2 * sn);
                f[3*i+a] -= de * gi; f[3*k+a] -= de * gk; f[3*j+a] += de * (gi + gk);
            }
            touch(tb, i); touch(tb, j); touch(tb, k);
        }
#pragma omp for schedule(static) nowait
        for (uint32_t t = 0; t < s->n_dihedrals; ++t) {
            const uint32_t i = s->dihedral_idx[4*t], j = s->dihedral_idx[4*t+1], k = s->dihedral_idx[4*t+2], l = s->dihedral_idx[4*t+3];
            float F[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
            float G[3] = { x[3*j]-x[3*k], x[3*j+1]-x[3*k+1], x[3*j+2]-x[3*k+2] };
            float H[3] = { x[3*l]-x[3*k], x[3*l+1]-x[3*k+1], x[3*l+2]-x[3*k+2] };
            mimg(p, F); mimg(p, G); mimg(p, H);
            const float A[3] = { F[1]*G[2]-F[2]*G[1], F[2]*G[0]-F[0]*G[2], F[0]*G[1]-F[1]*G[0] };
            const float B[3] = { H[1]*G[2]-H[2]*G[1], H[2]*G[0]-H[0]*G[2], H[0]*G[1]-H[1]*G[0] };
            const float A2 = A[0]*A[0]+A[1]*A[1]+A[2]*A[2], B2 = B[0]*B[0]+B[1]*B[1]+B[2]*B[2];
            const float Gn = sqrtf(G[0]*G[0]+G[1]*G[1]+G[2]*G[2]);
            if (A2 < 1e-12f || B2 < 1e-12f || Gn < 1e-6f) continue;
            const float cosphi = A[0]*B[0]+A[1]*B[1]+A[2]*B[2];
            const float BxA[3] = { B[1]*A[2]-B[2]*A[1], B[2]*A[0]-B[0]*A[2], B[0]*A[1]-B[1]*A[0] };
            const float sinphi = (BxA[0]*G[0]+BxA[1]*G[1]+BxA[2]*G[2]) / Gn;
            const float phi = atan2f(sinphi, cosphi), n = (float)s->dihedral_n[t], v = s->dihedral_v[t], ph = s->dihedral_phase[t];
            ed += (double)(v * (1.f + cosf(n * phi - ph)));
            const float de = -v * n * sinf(n * phi - ph);
            const float FG = F[0]*G[0]+F[1]*G[1]+F[2]*G[2], HG = H[0]*G[0]+H[1]*G[1]+H[2]*G[2];
            for (int a = 0; a < 3; ++a) {
                const float dpi = -Gn / A2 * A[a], dpl = Gn / B2 * B[a];
                const float dpj = Gn / A2 * A[a] + FG / (A2 * Gn) * A[a] - HG / (B2 * Gn) * B[a];
                const float dpk = -Gn / B2 * B[a] - FG / (A2 * Gn) * A[a] + HG / (B2 * Gn) * B[a];
                f[3*i+a] -= de * dpi; f[3*j+a] -= de * dpj; f[3*k+a] -= de * dpk; f[3*l+a] -= de * dpl;
            }
            touch(tb, i); touch(tb, j); touch(tb, k); touch(tb, l);
        }
#pragma omp for schedule(static)
        for (uint32_t q = 0; q < s->n_pairs14; ++q) {
            const uint32_t i = s->pairs14_idx[2*q], j = s->pairs14_idx[2*q+1];
            if (s->flags && ((s->flags[i] | s->flags[j]) & MDX_ATOM_BONDED_ONLY)) continue;
            float d[3] = { x[3*i]-x[3*j], x[3*i+1]-x[3*j+1], x[3*i+2]-x[3*j+2] };
            mimg(p, d);
            const float r2 = d[0]*d[0]+d[1]*d[1]+d[2]*d[2], inv_r = 1.f / sqrtf(r2), inv_r2 = inv_r * inv_r, r6 = inv_r2 * inv_r2 * inv_r2;
            const float c12 = c->scale14_lj * p->c12[s->lj_type[i] * T + s->lj_type[j]], c6 = c->scale14_lj * p->c6[s->lj_type[i] * T + s->lj_type[j]];
            const float qq = c->scale14_coulomb * p->qs[i] * p->qs[j];
            const float fs = (12.f * c12 * r6 - 6.f * c6) * r6 * inv_r2 + qq * inv_r * inv_r2;
            e14l += (double)((c12 * r6 - c6) * r6); e14c += (double)(qq * inv_r);
            for (int a = 0; a < 3; ++a) { f[3*i+a] += fs * d[a]; f[3*j+a] -= fs * d[a]; }
            touch(tb, i); touch(tb, j);
        }
    }
    if (want_e) { en[PE_BOND] += eb; en[PE_ANGLE] += ea; en[PE_DIHEDRAL] += ed; en[PE_LJ14] += e14l; en[PE_COUL14] += e14c; }
}

/* f = sum over the threads' private buffers (only the blocks a thread touched), buffers cleared on the way */
static void reduce_forces(prod_t* p, float* f) {
    const uint32_t N = p->N;
#pragma omp parallel for schedule(static)
    for (uint32_t blk = 0; blk < p->nblk; ++blk) {
        const uint32_t i0 = blk * BLK, i1 = (i0 + BLK < N) ? i0 + BLK : N;
        memset(f + 3 * (size_t)i0, 0, sizeof(float) * 3 * (i1 - i0));
        for (int t = 0; t < p->nthreads; ++t) {
            if (!p->touched[t][blk]) continue;
            float* src = p->fb[t] + 3 * (size_t)i0;
            for (uint32_t k = 0; k < 3 * (i1 - i0); ++k) { f[3 * (size_t)i0 + k] += src[k]; src[k] = 0.f; }
            p->touched[t][blk] = 0;
        }
    }
}

static void prod_forces(prod_t* p, const float* x, float* f, int want_e, double* en) {
    if (want_e) pair_rows(p, x, p->s->lj_type, 1, en); else pair_rows(p, x, p->s->lj_type, 0, en);
    bonded(p, x, want_e, en);
    reduce_forces(p, f);
}

static int list_stale(const prod_t* p, const float* x) {
    int stale = 0;
#pragma omp parallel for schedule(static) reduction(| : stale)
    for (uint32_t i = 0; i < p->N; ++i) {
        const float dx = x[3*i] - p->xref[3*i], dy = x[3*i+1] - p->xref[3*i+1], dz = x[3*i+2] - p->xref[3*i+2];
        stale |= !(dx * dx + dy * dy + dz * dz <= p->half_skin2);
    }
    return stale;
}

/* ---- entry points ---------------------------------------------------------------------------------------------- */
/* Forces and energies (en[PE_N]) of one configuration: list built for it. */
int cpu_prod_forces(const mdx_system* s, const mdx_config* c, const float* x, float* f, double* en) {
    prod_t* p = prod_create(s, c);
    if (!p) return -1;
    memset(en, 0, sizeof(double) * PE_N);
    prod_rebuild(p, x);
    prod_forces(p, x, f, 1, en);
    prod_destroy(p);
    return 0;
}

/* n_steps of velocity Verlet in place (x, v: [3N] fp32); the Verlet list is rebuilt when an atom has moved more than
 * skin/2 since the last build.  energy_every > 0: energies every that many steps (the last set is returned in en).
 * Returns the number of list builds, or -1. */
int cpu_prod_run(const mdx_system* s, const mdx_config* c, float* x, float* v, float dt, uint32_t n_steps,
                 uint32_t energy_every, double* en) {
    const int verbose = getenv("CPU_PROD_VERBOSE") != NULL;
    double t0 = omp_get_wtime();
    prod_t* p = prod_create(s, c);
    if (!p) return -1;
    const uint32_t N = p->N;
    float* f = (float*)malloc(sizeof(float) * 3 * (size_t)N);
    double e[PE_N]; memset(e, 0, sizeof(e));
    double t1 = omp_get_wtime();
    prod_rebuild(p, x);
    double t2 = omp_get_wtime();
    prod_forces(p, x, f, 0, e);
    if (verbose) fprintf(stderr, "cpu_prod: create %.3f s, first list build %.3f s, first force pass %.3f s\n", t1 - t0, t2 - t1, omp_get_wtime() - t2);
    for (uint32_t st = 0; st < n_steps; ++st) {
        const float hdt = 0.5f * dt;
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < N; ++i) {
            const float a = p->invm[i];
            for (int k = 0; k < 3; ++k) { v[3*i+k] += hdt * a * f[3*i+k]; x[3*i+k] += (a != 0.f ? dt : 0.f) * v[3*i+k]; }
        }
        if (list_stale(p, x)) prod_rebuild(p, x);
        const int want_e = energy_every && ((st + 1) % energy_every == 0 || st + 1 == n_steps);
        if (want_e) memset(e, 0, sizeof(e));
        prod_forces(p, x, f, want_e, e);
        double ke = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : ke)
        for (uint32_t i = 0; i < N; ++i) {
            const float a = p->invm[i];
            for (int k = 0; k < 3; ++k) { v[3*i+k] += hdt * a * f[3*i+k]; if (a != 0.f) ke += 0.5 * (double)(ACC_CONV / a) * v[3*i+k] * v[3*i+k] / ACC_CONV; }
        }
        if (want_e) e[PE_KIN] = ke;
    }
    if (en) memcpy(en, e, sizeof(e));
    g_last_pairs = p->pairs_evaluated;
    const int rb = (int)p->rebuilds;
    free(f);
    prod_destroy(p);
    return rb;
}
