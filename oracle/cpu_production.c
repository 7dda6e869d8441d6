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
 * steps until an atom has moved skin/2, OpenMP over all host cores, -O3 -march=native.  Round 4: the pair loop is a
 * CLUSTER-pair loop (the reference's CPU path is "thread pools and SIMD ... 512-bit, or 256-bit", README.md:208-211): atoms
 * are sorted into spatial clusters of 8 (structure-of-arrays coordinates per cluster), the list holds cluster pairs with
 * their periodic shift, and the inner loop evaluates one i-atom against the 8 j-atoms of a cluster per SIMD iteration - no
 * gathers, no per-pair minimum image, the j-forces of an entry accumulate in registers.  The atom-pair list of rounds 1-3
 * (gather per pair, 10 M list pairs/s/core) stays selectable with CPU_PROD_ATOM_LIST=1.  The fp64 oracle
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
    /* cluster-pair list (default): NS = 8 * NC slots; slot -> atom (UINT32_MAX = padding) */
    int use_clusters;
    uint32_t NC, NS; uint32_t cap_slots;
    uint32_t* c_atom;                      /* [NS] */
    float *cxs, *cys, *czs;                /* [NS] coordinates in the frame of the last rebuild (x - wrap offset) */
    float *cwx, *cwy, *cwz;                /* [NS] the wrap offset of the slot's atom at the last rebuild */
    float *cq, *csg, *cep;                 /* [NS] q sqrt(ke); sigma / 2 (or sqrt sigma); sqrt(4 eps) */
    uint64_t* cl_off; uint32_t* cl_ent; uint64_t cl_cap;    /* [NC + 1]; entries: J | shift code << 24 | masked << 31 */
    uint64_t* cl_moff; uint64_t* cl_mask; uint64_t cl_mcap; /* masks of the masked entries of i-cluster I start at cl_moff[I] */
    uint32_t* crow_lo;                     /* [nthreads + 1] i-clusters per thread, equal entry counts */
    float** cfb;                           /* per-thread force buffers in SLOT space, SoA: [3][cap_slots] */
    uint8_t** ctouched; uint32_t ncblk;    /* touched blocks of BLK slots */
    uint32_t *cblk_lo, *cblk_hi;
    uint64_t cluster_pair_evals;           /* 64 x cluster pairs walked by the force passes so far */
    uint64_t list_atom_pairs;              /* atom pairs inside the list radius the current cluster list covers */
} prod_t;

static int thread_count(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
int cpu_prod_max_threads(void) { return thread_count(); }
/* the host may be allowed fewer CPUs than it shows (a cgroup quota): bench.py sets the team to what it may really use */
void cpu_prod_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
static uint64_t g_last_pairs = 0, g_last_lane_pairs = 0;   /* list pairs (atom pairs inside the list radius) evaluated by the last cpu_prod_run (all force passes); lane pairs of the cluster loop */
uint64_t cpu_prod_last_pairs(void) { return g_last_pairs; }
uint64_t cpu_prod_last_lane_pairs(void) { return g_last_lane_pairs; }

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

static void bit_lut_init(void);
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
    p->use_clusters = getenv("CPU_PROD_ATOM_LIST") == NULL;
    bit_lut_init();
    p->crow_lo = (uint32_t*)calloc((size_t)p->nthreads + 1, sizeof(uint32_t));
    p->cblk_lo = (uint32_t*)calloc((size_t)p->nthreads, sizeof(uint32_t));
    p->cblk_hi = (uint32_t*)calloc((size_t)p->nthreads, sizeof(uint32_t));
    p->cfb = (float**)calloc(p->nthreads, sizeof(float*)); p->ctouched = (uint8_t**)calloc(p->nthreads, sizeof(uint8_t*));
    return p;
}

static void prod_destroy(prod_t* p) {
    if (!p) return;
    for (int t = 0; t < p->nthreads; ++t) { free(p->fb[t]); free(p->touched[t]); }
    free(p->fb); free(p->touched); free(p->c12); free(p->c6); free(p->qs); free(p->invm);
    for (int t = 0; t < p->nthreads; ++t) { free(p->cfb[t]); free(p->ctouched[t]); }
    free(p->cfb); free(p->ctouched); free(p->c_atom); free(p->cxs); free(p->cys); free(p->czs); free(p->cwx); free(p->cwy); free(p->cwz);
    free(p->cq); free(p->csg); free(p->cep); free(p->cl_off); free(p->cl_ent); free(p->cl_moff); free(p->cl_mask); free(p->crow_lo);
    free(p->cblk_lo); free(p->cblk_hi);
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


/* ================================================================================================================== */
/* Cluster-pair list and its SIMD pair loop (round 4)                                                                  */
/* ================================================================================================================== */
#define CL 8u
#define CL_NONE 0xFFFFFFFFu

/* bit k of the index -> lane k is 1.0f: an i-atom's row of an exclusion mask as a vector load */
static float BIT_LUT[256][CL];
static void bit_lut_init(void) { for (int m = 0; m < 256; ++m) for (int k = 0; k < 8; ++k) BIT_LUT[m][k] = ((m >> k) & 1) ? 1.f : 0.f; }

static void* xrealloc_f(void* p, size_t bytes) { free(p); return malloc(bytes ? bytes : 16); }

/* Atoms -> fine cells (edge ~ (8 / density)^(1/3): about one cluster per cell) sorted with x fastest; every (y, z) row of cells
 * is padded to whole clusters, so a cluster never runs from the end of one row into the start of the next.  Then the half
 * list of cluster pairs within the list radius (bounding boxes), every pair with its periodic shift; entries that hold an
 * excluded atom pair or pair a cluster with itself carry a 64-bit mask (bit 8 * ii + k: i-atom ii interacts with j-atom k). */
static void prod_rebuild_clusters(prod_t* p, const float* x) {
    const uint32_t N = p->N;
    const float rl = sqrtf(p->rl2);
    const double vol = (double)p->L[0] * p->L[1] * p->L[2];
    const float a0 = (float)cbrt(8.0 * vol / (double)N);
    int nc[3]; float w[3];
    for (int a = 0; a < 3; ++a) { nc[a] = (int)floorf(p->L[a] / a0); if (nc[a] < 3) nc[a] = 3; if (nc[a] > 1024) nc[a] = 1024; w[a] = p->L[a] / nc[a]; }
    const size_t ncell = (size_t)nc[0] * nc[1] * nc[2], nrow = (size_t)nc[1] * nc[2];
    uint32_t* start = (uint32_t*)calloc(ncell + 1, sizeof(uint32_t));
    uint32_t* cell = (uint32_t*)malloc(sizeof(uint32_t) * N);
    float* wrp = (float*)malloc(sizeof(float) * 3 * (size_t)N);
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < N; ++i) {
        int cc[3];
        for (int a = 0; a < 3; ++a) {
            const float off = floorf((x[3 * i + a] - p->lo[a]) * p->invL[a]) * p->L[a];
            wrp[3 * i + a] = off;
            int k = (int)((x[3 * i + a] - off - p->lo[a]) / w[a]); if (k < 0) k = 0; if (k >= nc[a]) k = nc[a] - 1;
            cc[a] = k;
        }
        cell[i] = (uint32_t)((cc[2] * nc[1] + cc[1]) * nc[0] + cc[0]);
    }
    for (uint32_t i = 0; i < N; ++i) start[cell[i] + 1]++;
    for (size_t k = 0; k < ncell; ++k) start[k + 1] += start[k];
    /* slots: rows padded to whole clusters */
    uint32_t* row_slot0 = (uint32_t*)malloc(sizeof(uint32_t) * (nrow + 1));
    uint32_t ns = 0;
    for (size_t r = 0; r < nrow; ++r) {
        row_slot0[r] = ns;
        const uint32_t n_row = start[(r + 1) * nc[0]] - start[r * nc[0]];
        ns += (n_row + CL - 1) / CL * CL;
    }
    row_slot0[nrow] = ns;
    p->NS = ns; p->NC = ns / CL;
    if (ns > p->cap_slots) {
        p->cap_slots = ns + ns / 16 + 64;
        const size_t b = sizeof(float) * p->cap_slots;
        p->c_atom = (uint32_t*)xrealloc_f(p->c_atom, sizeof(uint32_t) * p->cap_slots);
        p->cxs = (float*)xrealloc_f(p->cxs, b); p->cys = (float*)xrealloc_f(p->cys, b); p->czs = (float*)xrealloc_f(p->czs, b);
        p->cwx = (float*)xrealloc_f(p->cwx, b); p->cwy = (float*)xrealloc_f(p->cwy, b); p->cwz = (float*)xrealloc_f(p->cwz, b);
        p->cq = (float*)xrealloc_f(p->cq, b); p->csg = (float*)xrealloc_f(p->csg, b); p->cep = (float*)xrealloc_f(p->cep, b);
        p->cl_off = (uint64_t*)xrealloc_f(p->cl_off, sizeof(uint64_t) * (p->cap_slots / CL + 2));
        p->cl_moff = (uint64_t*)xrealloc_f(p->cl_moff, sizeof(uint64_t) * (p->cap_slots / CL + 2));
        p->ncblk = (p->cap_slots + BLK - 1) / BLK;
        for (int t = 0; t < p->nthreads; ++t) {
            free(p->cfb[t]); free(p->ctouched[t]);
            p->cfb[t] = (float*)calloc((size_t)3 * p->cap_slots, sizeof(float));
            p->ctouched[t] = (uint8_t*)calloc(p->ncblk, 1);
        }
    }
    for (uint32_t s = 0; s < ns; ++s) p->c_atom[s] = CL_NONE;
    {
        uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * (ncell + 1));
        memcpy(cur, start, sizeof(uint32_t) * (ncell + 1));
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t c = cell[i];
            const size_t r = c / (uint32_t)nc[0];
            const uint32_t pos = cur[c]++;                                   /* position in the cell-sorted order */
            p->c_atom[row_slot0[r] + (pos - start[r * nc[0]])] = i;
        }
        free(cur);
    }
    const mdx_system* sy = p->s; const mdx_config* cf = p->c;
    const int geom = cf->combining_rule == MDX_COMBINE_GEOMETRIC, lj_off = (cf->overrides & MDX_OVR_LJ_DISABLED) != 0;
    uint32_t* slot_of = (uint32_t*)malloc(sizeof(uint32_t) * N);
#pragma omp parallel for schedule(static)
    for (uint32_t s = 0; s < ns; ++s) {
        const uint32_t i = p->c_atom[s];
        if (i == CL_NONE) {      /* padding: no charge, no LJ, parked far away at its own coordinate */
            p->cwx[s] = p->cwy[s] = p->cwz[s] = 0.f; p->cq[s] = 0.f; p->csg[s] = 0.f; p->cep[s] = 0.f;
            p->cxs[s] = 1.0e6f + 64.0f * (float)(s & 0xFFFF); p->cys[s] = 2.0e6f + 64.0f * (float)(s >> 16); p->czs[s] = 3.0e6f;
            continue;
        }
        slot_of[i] = s;
        p->cwx[s] = wrp[3 * i]; p->cwy[s] = wrp[3 * i + 1]; p->cwz[s] = wrp[3 * i + 2];
        p->cxs[s] = x[3 * i] - p->cwx[s]; p->cys[s] = x[3 * i + 1] - p->cwy[s]; p->czs[s] = x[3 * i + 2] - p->cwz[s];
        const uint8_t fl = sy->flags ? sy->flags[i] : 0;
        const float sg = sy->lj_sigma[sy->lj_type[i]], ep = (lj_off || (fl & MDX_ATOM_BONDED_ONLY)) ? 0.f : sy->lj_eps[sy->lj_type[i]];
        p->cq[s] = p->qs[i];
        p->csg[s] = geom ? sqrtf(sg) : 0.5f * sg;
        p->cep[s] = sqrtf(4.0f * ep);
    }
    /* cluster bounding boxes (real atoms only) */
    const uint32_t NC = p->NC;
    float* bb = (float*)malloc(sizeof(float) * 6 * (size_t)NC);
#pragma omp parallel for schedule(static)
    for (uint32_t c = 0; c < NC; ++c) {
        float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
        for (uint32_t k = 0; k < CL; ++k) {
            const uint32_t s = c * CL + k;
            if (p->c_atom[s] == CL_NONE) continue;
            const float v[3] = { p->cxs[s], p->cys[s], p->czs[s] };
            for (int a = 0; a < 3; ++a) { if (v[a] < lo[a]) lo[a] = v[a]; if (v[a] > hi[a]) hi[a] = v[a]; }
        }
        for (int a = 0; a < 3; ++a) { bb[6 * c + a] = lo[a]; bb[6 * c + 3 + a] = hi[a]; }
    }
    /* the search: two passes (count, fill) over the rows of cells a cluster's box reaches */
    const float rl2 = p->rl2;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            uint64_t te = 0, tm = 0;
            for (uint32_t c = 0; c < NC; ++c) { const uint64_t ne = p->cl_off[c], nm = p->cl_moff[c]; p->cl_off[c] = te; p->cl_moff[c] = tm; te += ne; tm += nm; }
            p->cl_off[NC] = te; p->cl_moff[NC] = tm;
            if (te > p->cl_cap) { p->cl_cap = te + te / 8 + 1024; p->cl_ent = (uint32_t*)xrealloc_f(p->cl_ent, sizeof(uint32_t) * p->cl_cap); }
            if (tm > p->cl_mcap) { p->cl_mcap = tm + tm / 8 + 1024; p->cl_mask = (uint64_t*)xrealloc_f(p->cl_mask, sizeof(uint64_t) * p->cl_mcap); }
        }
#pragma omp parallel for schedule(dynamic, 64)
        for (uint32_t I = 0; I < NC; ++I) {
            const float* bi = bb + 6 * (size_t)I;
            uint64_t ne = 0, nm = 0;
            uint32_t* out = pass ? p->cl_ent + p->cl_off[I] : NULL;
            uint64_t* mout = pass ? p->cl_mask + p->cl_moff[I] : NULL;
            if (bi[0] > bi[3]) { if (!pass) { p->cl_off[I] = 0; p->cl_moff[I] = 0; } continue; }      /* all padding */
            /* clusters that hold an excluded partner of one of this cluster's atoms */
            uint32_t xc[96]; int nxc = 0;
            for (uint32_t k = 0; k < CL; ++k) {
                const uint32_t i = p->c_atom[I * CL + k];
                if (i == CL_NONE) continue;
                for (uint32_t e = p->ex_off[i]; e < p->ex_off[i + 1]; ++e) {
                    const uint32_t cj = slot_of[p->ex_idx[e]] / CL;
                    int seen = 0;
                    for (int q = 0; q < nxc; ++q) if (xc[q] == cj) { seen = 1; break; }
                    if (!seen && nxc < 96) xc[nxc++] = cj;
                }
            }
            int c0[3], c1[3];
            for (int a = 0; a < 3; ++a) {
                c0[a] = (int)floorf((bi[a] - rl - p->lo[a]) / w[a]); c1[a] = (int)floorf((bi[3 + a] + rl - p->lo[a]) / w[a]);
                /* (a reach wider than the axis lists a cell under two images: the bounding-box test keeps the ones in range, and an atom
                 * pair is inside the cutoff under one image only - the box is at least 2 r_list) */
                if (c0[a] < -nc[a]) c0[a] = -nc[a];
                if (c1[a] > 2 * nc[a] - 1) c1[a] = 2 * nc[a] - 1;
            }
            for (int gz = c0[2]; gz <= c1[2]; ++gz) {
                const int kz = (gz >= 0 ? gz / nc[2] : -((-gz + nc[2] - 1) / nc[2])), az = gz - kz * nc[2];
                const float sz = (float)kz * p->L[2];
                for (int gy = c0[1]; gy <= c1[1]; ++gy) {
                    const int ky = (gy >= 0 ? gy / nc[1] : -((-gy + nc[1] - 1) / nc[1])), ay = gy - ky * nc[1];
                    const float sy2 = (float)ky * p->L[1];
                    const size_t r = (size_t)az * nc[1] + ay;
                    /* the row's x range, split where it wraps */
                    for (int seg = 0; seg < 3; ++seg) {
                        const int kx = seg - 1;
                        int x0 = c0[0] > kx * nc[0] ? c0[0] : kx * nc[0], x1 = c1[0] < (kx + 1) * nc[0] - 1 ? c1[0] : (kx + 1) * nc[0] - 1;
                        if (x0 > x1) continue;
                        x0 -= kx * nc[0]; x1 -= kx * nc[0];
                        const float sx = (float)kx * p->L[0];
                        const uint32_t a_lo = start[r * nc[0] + x0] - start[r * nc[0]], a_hi = start[r * nc[0] + x1 + 1] - start[r * nc[0]];
                        if (a_hi <= a_lo) continue;
                        const uint32_t J0 = (row_slot0[r] + a_lo) / CL, J1 = (row_slot0[r] + a_hi - 1) / CL;
                        const uint32_t code = (uint32_t)((kx + 1) + 3 * (ky + 1) + 9 * (kz + 1));
                        for (uint32_t J = J0; J <= J1; ++J) {
                            if (J < I) continue;                                                        /* half list: J >= I */
                            if (J == I && code != 13u) { if (code < 13u) continue; }                   /* a cluster's pair with its own image: once */
                            const float* bj = bb + 6 * (size_t)J;
                            if (bj[0] > bj[3]) continue;
                            float d2 = 0.f;
                            const float sh[3] = { sx, sy2, sz };
                            for (int a = 0; a < 3; ++a) {
                                const float lo_j = bj[a] + sh[a], hi_j = bj[3 + a] + sh[a];
                                const float g = fmaxf(0.f, fmaxf(bi[a] - hi_j, lo_j - bi[3 + a]));
                                d2 += g * g;
                            }
                            if (!(d2 < rl2)) continue;
                            int masked = (J == I && code == 13u);
                            /* (whatever the image: atoms are wrapped one by one, so a molecule across a face has its excluded pairs in
                             * an entry with a shift; the box is > 2 r_list, so a pair is in range under one image only) */
                            for (int q = 0; q < nxc && !masked; ++q) if (xc[q] == J) masked = 1;
                            if (pass) {
                                out[ne] = J | (code << 24) | ((uint32_t)masked << 31);
                                if (masked) {
                                    uint64_t m = 0;
                                    for (uint32_t ii = 0; ii < CL; ++ii) {
                                        const uint32_t ia = p->c_atom[I * CL + ii];
                                        for (uint32_t k = 0; k < CL; ++k) {
                                            const uint32_t ja = p->c_atom[J * CL + k];
                                            int ok = ia != CL_NONE && ja != CL_NONE;
                                            if (ok && J == I && code == 13u) ok = k > ii;       /* (its pair with its own image: all 64) */
                                            if (ok && excluded(p, ia, ja)) ok = 0;
                                            if (ok) m |= 1ull << (8 * ii + k);
                                        }
                                    }
                                    mout[nm] = m;
                                }
                            }
                            ++ne; nm += masked;
                        }
                    }
                }
            }
            if (!pass) { p->cl_off[I] = ne; p->cl_moff[I] = nm; }
        }
    }
    {   /* i-clusters -> threads by equal entry counts; the window of slot blocks a thread writes to */
        const uint64_t tot = p->cl_off[NC];
        uint32_t r = 0;
        for (int t = 0; t < p->nthreads; ++t) {
            p->crow_lo[t] = r;
            const uint64_t goal = tot * (uint64_t)(t + 1) / (uint64_t)p->nthreads;
            while (r < NC && p->cl_off[r + 1] <= goal) ++r;
            if (t == p->nthreads - 1) r = NC;
        }
        p->crow_lo[p->nthreads] = NC;
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < p->nthreads; ++t) {
            const uint32_t a = p->crow_lo[t], b = p->crow_lo[t + 1];
            uint32_t mx = b ? b - 1 : 0, mn = a;
            for (uint64_t k = p->cl_off[a]; k < p->cl_off[b]; ++k) { const uint32_t J = p->cl_ent[k] & 0xFFFFFFu; if (J > mx) mx = J; if (J < mn) mn = J; }
            p->cblk_lo[t] = (mn * CL) / BLK; p->cblk_hi[t] = (a < b) ? (mx * CL + CL - 1) / BLK + 1 : (mn * CL) / BLK;
        }
    }
    memcpy(p->xref, x, sizeof(float) * 3 * (size_t)N);
    free(start); free(cell); free(wrp); free(row_slot0); free(slot_of); free(bb);
    p->rebuilds++;
}

/* slot-space coordinates of the current step (the atoms keep the frame they had at the rebuild) */
static void clusters_refresh(prod_t* p, const float* x) {
#pragma omp parallel for schedule(static)
    for (uint32_t s = 0; s < p->NS; ++s) {
        const uint32_t i = p->c_atom[s];
        if (i == CL_NONE) continue;
        p->cxs[s] = x[3 * i] - p->cwx[s]; p->cys[s] = x[3 * i + 1] - p->cwy[s]; p->czs[s] = x[3 * i + 2] - p->cwz[s];
    }
}

/* One i-cluster against its list.  The eight i-atoms are walked one at a time against the eight j-atoms of an entry: every
 * statement of the k loop is an 8-lane vector operation on contiguous data (a ymm register; zmm holds two i-atoms' worth when
 * the compiler interleaves), the j-forces of the entry stay in registers until all eight i-atoms have been through. */
static inline __attribute__((always_inline)) void pair_clusters(prod_t* p, const int want_e, const int has_soft, double* en) {
    const mdx_config* c = p->c;
    const float rc2l = p->rc2_lj, rc2c = p->rc2_coul, soft = c->softening_sq;
    const int rf = c->coulomb_mode == MDX_COULOMB_REACTION, geom = c->combining_rule == MDX_COMBINE_GEOMETRIC;
    const float rc = c->coulomb_cutoff;
    const float krf2 = rf ? 1.0f / (rc * rc * rc) : 0.f, krf = 0.5f * krf2, crf = rf ? 1.5f / rc : 1.0f / rc;
    const size_t cap = p->cap_slots;
    double e_lj = 0.0, e_c = 0.0;
#pragma omp parallel reduction(+ : e_lj, e_c)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), team = omp_get_num_threads();
#else
        const int tid = 0, team = 1;
#endif
        float* fbx = p->cfb[tid]; float* fby = fbx + cap; float* fbz = fby + cap;
        uint8_t* tb = p->ctouched[tid];
        for (int share = tid; share < p->nthreads; share += team) {
            for (uint32_t bbk = p->cblk_lo[share]; bbk < p->cblk_hi[share]; ++bbk) tb[bbk] = 1;
            for (uint32_t I = p->crow_lo[share]; I < p->crow_lo[share + 1]; ++I) {
                const uint64_t ea = p->cl_off[I], eb = p->cl_off[I + 1];
                if (ea == eb) continue;
                const float* xi = p->cxs + (size_t)I * CL; const float* yi = p->cys + (size_t)I * CL; const float* zi = p->czs + (size_t)I * CL;
                const float* qi = p->cq + (size_t)I * CL; const float* sgi = p->csg + (size_t)I * CL; const float* epi = p->cep + (size_t)I * CL;
                /* the force on i-atom ii accumulates LANE-WISE over the j-atoms of every entry (8 x 3 vector accumulators) and is
                 * summed across the lanes once per i-cluster: no horizontal add inside the entry loop */
                float fix[CL][CL], fiy[CL][CL], fiz[CL][CL];
                for (uint32_t ii = 0; ii < CL; ++ii)
#pragma omp simd
                    for (uint32_t k = 0; k < CL; ++k) { fix[ii][k] = 0.f; fiy[ii][k] = 0.f; fiz[ii][k] = 0.f; }
                float elj = 0.f, ec = 0.f;
                const uint64_t* mk = p->cl_mask + p->cl_moff[I];
                for (uint64_t e = ea; e < eb; ++e) {
                    const uint32_t ent = p->cl_ent[e];
                    const uint32_t J = ent & 0xFFFFFFu, code = (ent >> 24) & 31u;
                    const uint64_t mask = (ent >> 31) ? *mk++ : ~0ull;
                    const float shx = (float)((int)(code % 3u) - 1) * p->L[0], shy = (float)((int)((code / 3u) % 3u) - 1) * p->L[1],
                                shz = (float)((int)(code / 9u) - 1) * p->L[2];
                    float xj[CL], yj[CL], zj[CL], qj[CL], sgj[CL], epj[CL], fjx[CL], fjy[CL], fjz[CL];
#pragma omp simd
                    for (uint32_t k = 0; k < CL; ++k) {
                        xj[k] = p->cxs[(size_t)J * CL + k] + shx; yj[k] = p->cys[(size_t)J * CL + k] + shy; zj[k] = p->czs[(size_t)J * CL + k] + shz;
                        qj[k] = p->cq[(size_t)J * CL + k]; sgj[k] = p->csg[(size_t)J * CL + k]; epj[k] = p->cep[(size_t)J * CL + k];
                        fjx[k] = 0.f; fjy[k] = 0.f; fjz[k] = 0.f;
                    }
#define PAIR_ROW(ALLOW)                                                                                                       \
                    for (uint32_t ii = 0; ii < CL; ++ii) {                                                               \
                        const float xa = xi[ii], ya = yi[ii], za = zi[ii], qa = qi[ii], sga = sgi[ii], epa = epi[ii];    \
                        const float* const al = BIT_LUT[(uint32_t)(mask >> (8 * ii)) & 0xFFu];                           \
                        _Pragma("omp simd reduction(+ : elj, ec)")                                                       \
                        for (uint32_t k = 0; k < CL; ++k) {                                                              \
                            const float dx = xa - xj[k], dy = ya - yj[k], dz = za - zj[k];                               \
                            const float r2 = dx * dx + dy * dy + dz * dz;                                                \
                            const float allow = (ALLOW) ? al[k] : 1.f;                                                   \
                            const float ml = (r2 < rc2l ? 1.f : 0.f) * allow, mc = (r2 < rc2c ? 1.f : 0.f) * allow;      \
                            /* (a masked-out self pair has r = 0: keep 0 x inf out) */                                   \
                            const float r2s = (ALLOW) ? ((allow != 0.f && r2 > 1e-12f) ? r2 : 1.0f) : r2;                \
                            const float inv_r = 1.0f / sqrtf(r2s), inv_r2 = inv_r * inv_r;                               \
                            const float sg = geom ? sga * sgj[k] : sga + sgj[k];                                         \
                            const float e4 = epa * epj[k] * ml;                           /* 4 eps_ij */                 \
                            const float s2 = sg * sg * inv_r2, s6 = s2 * s2 * s2;                                        \
                            float fs = 6.0f * e4 * s6 * (2.0f * s6 - 1.0f) * inv_r2;      /* 24 eps (2 s12 - s6) / r^2 */ \
                            const float qq = qa * qj[k] * mc;                                                            \
                            if (has_soft) fs += rf ? qq * (inv_r * inv_r2 - krf2) : qq * inv_r / (r2s + soft);           \
                            else fs += qq * (inv_r * inv_r2 - krf2);                      /* (krf2 = 0 without reaction field) */ \
                            if (want_e) {                                                                                \
                                elj += e4 * s6 * (s6 - 1.0f);                                                            \
                                ec += rf ? qq * (inv_r + krf * r2 - crf) : qq * (inv_r - crf);                           \
                            }                                                                                            \
                            const float gx = fs * dx, gy = fs * dy, gz = fs * dz;                                        \
                            fix[ii][k] += gx; fiy[ii][k] += gy; fiz[ii][k] += gz;                                        \
                            fjx[k] -= gx; fjy[k] -= gy; fjz[k] -= gz;                                                    \
                        }                                                                                                \
                    }
                    /* (entries without an exclusion or a self pair - nine in ten - run a body without the mask) */
                    if (ent >> 31) { PAIR_ROW(1) } else { PAIR_ROW(0) }
#undef PAIR_ROW
#pragma omp simd
                    for (uint32_t k = 0; k < CL; ++k) { fbx[(size_t)J * CL + k] += fjx[k]; fby[(size_t)J * CL + k] += fjy[k]; fbz[(size_t)J * CL + k] += fjz[k]; }
                }
                for (uint32_t ii = 0; ii < CL; ++ii) {
                    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma omp simd reduction(+ : sx, sy, sz)
                    for (uint32_t k = 0; k < CL; ++k) { sx += fix[ii][k]; sy += fiy[ii][k]; sz += fiz[ii][k]; }
                    fbx[(size_t)I * CL + ii] += sx; fby[(size_t)I * CL + ii] += sy; fbz[(size_t)I * CL + ii] += sz;
                }
                e_lj += elj; e_c += ec;
            }
        }
    }
    p->cluster_pair_evals += 64ull * p->cl_off[p->NC];
    if (want_e) { en[PE_LJ] += e_lj; en[PE_COUL] += e_c; }
}

/* the threads' slot-space buffers -> f (atom space), which reduce_forces has just filled with the bonded forces */
static void reduce_cluster_forces(prod_t* p, float* f) {
    const size_t cap = p->cap_slots;
#pragma omp parallel for schedule(static)
    for (uint32_t blk = 0; blk < (p->NS + BLK - 1) / BLK; ++blk) {
        const uint32_t s0 = blk * BLK, s1 = (s0 + BLK < p->NS) ? s0 + BLK : p->NS;
        for (int t = 0; t < p->nthreads; ++t) {
            if (!p->ctouched[t][blk]) continue;
            float* bx = p->cfb[t]; float* by = bx + cap; float* bz = by + cap;
            for (uint32_t s = s0; s < s1; ++s) {
                const uint32_t i = p->c_atom[s];
                if (i != CL_NONE) { f[3 * (size_t)i] += bx[s]; f[3 * (size_t)i + 1] += by[s]; f[3 * (size_t)i + 2] += bz[s]; }
                bx[s] = 0.f; by[s] = 0.f; bz[s] = 0.f;
            }
            p->ctouched[t][blk] = 0;
        }
    }
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
    if (p->use_clusters) {
        clusters_refresh(p, x);
        const int soft = p->c->softening_sq != 0.f;
        if (want_e) { if (soft) pair_clusters(p, 1, 1, en); else pair_clusters(p, 1, 0, en); }
        else { if (soft) pair_clusters(p, 0, 1, en); else pair_clusters(p, 0, 0, en); }
        /* (the figure bench.py quotes counts ATOM pairs inside the list radius, as the atom list did: ~0.35 of the lane pairs) */
        p->pairs_evaluated += p->list_atom_pairs;
    } else {
        if (want_e) pair_rows(p, x, p->s->lj_type, 1, en); else pair_rows(p, x, p->s->lj_type, 0, en);
    }
    bonded(p, x, want_e, en);
    reduce_forces(p, f);
    if (p->use_clusters) reduce_cluster_forces(p, f);
}

static void prod_rebuild_any(prod_t* p, const float* x) {
    if (!p->use_clusters) { prod_rebuild(p, x); return; }
    prod_rebuild_clusters(p, x);
    /* atom pairs inside the list radius covered by the cluster list (statistics: the unit of list_pairs_per_s) */
    uint64_t n = 0;
    const float rl2 = p->rl2;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : n)
    for (uint32_t I = 0; I < p->NC; ++I) {
        const uint64_t* mk = p->cl_mask + p->cl_moff[I];
        for (uint64_t e = p->cl_off[I]; e < p->cl_off[I + 1]; ++e) {
            const uint32_t ent = p->cl_ent[e], J = ent & 0xFFFFFFu, code = (ent >> 24) & 31u;
            const uint64_t mask = (ent >> 31) ? *mk++ : ~0ull;
            const float shx = (float)((int)(code % 3u) - 1) * p->L[0], shy = (float)((int)((code / 3u) % 3u) - 1) * p->L[1], shz = (float)((int)(code / 9u) - 1) * p->L[2];
            for (uint32_t ii = 0; ii < CL; ++ii)
                for (uint32_t k = 0; k < CL; ++k) {
                    if (!((mask >> (8 * ii + k)) & 1ull)) continue;
                    const float dx = p->cxs[I * CL + ii] - p->cxs[J * CL + k] - shx, dy = p->cys[I * CL + ii] - p->cys[J * CL + k] - shy, dz = p->czs[I * CL + ii] - p->czs[J * CL + k] - shz;
                    n += dx * dx + dy * dy + dz * dz < rl2;
                }
        }
    }
    p->list_atom_pairs = n;
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
    prod_rebuild_any(p, x);
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
    prod_rebuild_any(p, x);
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
        if (list_stale(p, x)) prod_rebuild_any(p, x);
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
    g_last_pairs = p->pairs_evaluated; g_last_lane_pairs = p->cluster_pair_evals;
    const int rb = (int)p->rebuilds;
    free(f);
    prod_destroy(p);
    return rb;
}
