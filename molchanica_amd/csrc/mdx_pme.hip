// mdx_pme.hip — smooth particle-mesh Ewald reciprocal space.
//
// The reference's Coulomb is SPME unless `long_range_recip_disabled` (/root/reference README.md:240,
// src/mol_editor/mod.rs:873) and refuses the GPU path without an FFT library it dlopens
// (src/util.rs:1094-1100; make_release.sh:27-48 asserts no load-time CUDA libs); the implementation
// lives in the absent `ewald` crate.  Built here (Essmann et al., J. Chem. Phys. 103, 8577, 1995):
//   spread   one lane per slot: cubic B-spline weights (order 4), 64 f32 atomic adds into the
//            charge mesh.  Slots are spatially sorted, so a wave's adds land in a few L2 lines.
//   fft      hipFFT R2C / C2R, dlopen'ed on first use exactly like the reference does with cuFFT;
//            the library itself never links an FFT library.
//   solve    one lane per complex mesh point: F *= theta(m), theta = B(m) exp(-pi^2 m^2/beta^2)/(pi V m^2)
//            precomputed on the host in fp64; energy 1/2 sum theta |F|^2 (with the R2C multiplicity).
//   gather   one lane per slot: 64 reads of the potential mesh x spline derivatives -> force, added
//            to force[slot] (single writer).
// Charges on the mesh already carry sqrt(k_e) (posq.w), so energies and forces come out in kcal/mol.
// Self term, neutralising background and the erf(beta r)/r of excluded / 1-4 pairs (an extra role
// kind in the bonded gather) complete the Ewald sum.  All mesh kernels are HBM/atomic-bound.
#include "mdx_comm.h"
#include <climits>
#include <cstdlib>
#include <hipfft/hipfft.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

#define FAIL(code, msg) do { mdx_set_error(msg); return (code); } while (0)
static inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

struct PmeDev {
    float lo[3], inv_len[3], scale[3];   // scale = K / L
    int K[3];
    float fix, unfix;                    // fixed-point scale of the LDS charge canvases and its inverse (PME_FIX below)
};

struct PmeBrickGeom {
    int nb[3];                             // bricks per dimension; brick i of dimension d starts at (i * K[d]) / nb[d]
    int cb[3];                             // canvas extent = largest brick of the dimension + 3
    uint32_t nbricks, cap, stride;         // bucket capacity (records), canvas stride in scratch (floats)
    // brick of a mesh cell and the cell's place in it, per dimension: ctab[coff[d] + k] = brick << 16 | k - start(brick); and, for the
    // combine pass, the one or two canvas columns that hold mesh plane kz: ztab[kz] = (i0, i1 or -1).  (Computed per atom / per point
    // these were six and twelve 32-bit integer divisions by run-time values: half of the bin pass's instructions.)
    const uint32_t* ctab; int coff[3]; const int2* ztab;
};

struct PmePlan {
    void* lib = nullptr;
    decltype(&hipfftPlan3d) plan3d = nullptr;
    decltype(&hipfftExecR2C) exec_r2c = nullptr;
    decltype(&hipfftExecC2R) exec_c2r = nullptr;
    decltype(&hipfftSetStream) set_stream = nullptr;
    decltype(&hipfftDestroy) destroy = nullptr;
    hipfftHandle fwd{}, inv{};
    bool have_plans = false;
    // ---- hand-written x pass (single-device handles; see pme_xpass_solve_kernel) ----
    hipfftHandle fwd2{}, inv2{};           // K0 batches of the 2-D transform over (y, z)
    bool have_xpass = false;
    float2* tw = nullptr;                  // exp(-2 pi i k / K0), k < K0 (fp64 on the host)
    float* phi = nullptr;                  // the potential mesh of handles whose spread needs a cleared charge mesh (see XpassArgs::zero)
    int nfac = 0, fac[16] = {};
    PmeDev dev{};
    size_t n_real = 0, n_cplx = 0;      // n_cplx = K0 K1 pitch
    int pitch = 0;                      // complex numbers per (x, y) row of the half-complex mesh: K2 / 2 + 1, padded on single-device handles
    // ---- brick spread (single-GPU handles; see "Brick spread" below) ----
    struct Brick {
        bool on = false;
        PmeBrickGeom g{};
        float4* rec = nullptr; uint32_t* code = nullptr; uint32_t* slot = nullptr; uint32_t* count = nullptr; uint32_t* ovf = nullptr; int* scratch = nullptr;
        uint32_t* ctab = nullptr; int2* ztab = nullptr;
        uint64_t force_zero_epoch = ~0ull;      // list generation the separate force array was last cleared for (brick gather)
        uint32_t ovf_S = 0;
    } brick;
    // ---- slab-decomposed mesh of a decomposed handle (round 3; see "Slab-decomposed SPME" below) ----
    decltype(&hipfftPlanMany) plan_many = nullptr;
    decltype(&hipfftExecC2C) exec_c2c = nullptr;
    struct Group { int peer; int a_lo, a_cnt; uint32_t row0, nrows; };     // x-planes [a_lo, a_lo + a_cnt) of a block, for one slab owner
    struct Slab {
        bool on = false;
        int W = 1, rank = 0, nx = 0, ny = 0;                 // planes / rows of the mesh per rank
        int b0[32][3], nb[32][3];                            // every rank's spread block: origin (unwrapped mesh index) and extent
        std::vector<Group> out_groups;                       // my block, cut by slab owner (ascending a)
        std::vector<Group> in_groups;                        // the blocks of all ranks, the part inside my slab (by source rank, ascending a)
        std::vector<MdxSeg> s_out, r_in, s_tr, r_tr;         // segments of the redistribution (charges out; potential back = reversed) and of the transposes
        float *slab_real = nullptr; float2 *slab_cplx = nullptr, *tr = nullptr;
        float4 *buf_a = nullptr, *buf_b = nullptr; size_t cap_a = 0, cap_b = 0;    // staging of the redistribution / transposes
        hipfftHandle fwd2d{}, inv2d{}, fft1d{};
        bool have_plans = false;
        size_t max_out = 0, max_in = 0;                     // largest group (cells) of either list
    } slab;
};

__device__ __forceinline__ void bspline4(float w, float* m, float* d) {
    // row j belongs to mesh point floor(u) - 3 + j:  M4(w + 3 - j) and its derivative
    const float w2 = w * w, w3 = w2 * w, o = 1.0f - w;
    m[0] = o * o * o * (1.0f / 6.0f);
    m[1] = (3.0f * w3 - 6.0f * w2 + 4.0f) * (1.0f / 6.0f);
    m[2] = (-3.0f * w3 + 3.0f * w2 + 3.0f * w + 1.0f) * (1.0f / 6.0f);
    m[3] = w3 * (1.0f / 6.0f);
    d[0] = -0.5f * o * o;
    d[1] = 0.5f * (3.0f * w2 - 4.0f * w);
    d[2] = 0.5f * (-3.0f * w2 + 2.0f * w + 1.0f);
    d[3] = 0.5f * w2;
}

__device__ __forceinline__ void mesh_coords(const float4 p, const PmeDev& g, int* k0, float* w) {
    const float x[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float u = (x[a] - g.lo[a]) * g.inv_len[a];
        u = (u - floorf(u)) * (float)g.K[a];
        int fl = (int)floorf(u);
        if (fl >= g.K[a]) fl = g.K[a] - 1;
        w[a] = u - (float)fl;
        k0[a] = fl - 3;
    }
}

// 16 lanes per atom: lane (b, c) owns the y/z offsets and walks the 4 x offsets, so the 4 lanes of a
// c-group add to 4 consecutive floats (one 16-B segment): the adds of a wave fall into far fewer L2
// requests than one-lane-per-atom (1.58 ms -> see profiles/ at 1 M atoms, 240^3 mesh).
__global__ __launch_bounds__(256) void pme_spread_kernel(uint32_t S, const float4* __restrict__ posq,
                                                         const uint8_t* __restrict__ slot_flags, PmeDev g,
                                                         float* __restrict__ Q, const uint32_t* gate, uint32_t thr, uint32_t need,
                                                         const float2* __restrict__ lj, int sel) {
    if (gate && *gate > thr) return;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = t >> 4;
    const int b = (t >> 2) & 3, c = t & 3;
    if (s >= S) return;
    if ((slot_flags[s] & need) != need) return;    // need = 1: every real atom; 3: the atoms this rank owns (decomposed handle)
    // alchemical window: sel 1 = the environment only, 2 = the coupled molecule only (its atoms carry a negative sqrt(24 eps))
    if (sel && ((__float_as_int(lj[s].y) < 0) != (sel == 2))) return;
    const float4 p = posq[s];
    if (p.w == 0.f) return;
    int k0[3]; float w[3];
    mesh_coords(p, g, k0, w);
    float mx[4], my[4], mz[4], dd[4];
    bspline4(w[0], mx, dd); bspline4(w[1], my, dd); bspline4(w[2], mz, dd);
    int ky = k0[1] + b; if (ky < 0) ky += g.K[1];
    int kz = k0[2] + c; if (kz < 0) kz += g.K[2];
    const float qbc = p.w * my[b] * mz[c];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int kx = k0[0] + a; if (kx < 0) kx += g.K[0];
        atomicAdd(Q + ((size_t)kx * g.K[1] + ky) * g.K[2] + kz, qbc * mx[a]);
    }
}

// Tile-local spread (default).  The 64 atoms of a tile sit in an ~8.6 A brick, i.e. they touch a block of about
// 13^3 mesh points: one workgroup per tile accumulates the tile's 64 x 64 contributions in an LDS block (fixed point,
// see PME_FIX below) and flushes each touched point ONCE with a global atomic - ~2200 memory-side atomics per tile instead
// of 4096, in rows of consecutive floats.  (The memory side retires ~250 G f32 atomics/s: the 66 M adds of the
// per-atom kernel above cannot take less than 0.26 ms at 1 M atoms, and took 0.68.)  Mesh indices are kept
// unwrapped inside the block - an atom that has drifted across the box face since the last rebuild is still a
// neighbour of the rest of its tile - and wrapped at the flush; an atom whose footprint does not fit the block
// (a sparse tile) falls back to direct atomics.
// gfx950 retires ds_add_f32 at ONE lane per three clocks and CU - 0.33 lane-adds per clock whatever the addresses, against 10 per
// clock for ds_add_u32 (tools/ubench/lds_atomic_rate.hip, profiles/r04_lds_atomic_rate_ubench.txt): the LDS canvases of the charge
// spread therefore accumulate in 32-bit fixed point, and - integer adds commute - the sum no longer depends on the order of the lanes.
// The values spread are posq.w = q sqrt(k_e) (18.2 per elementary charge), times three spline weights (at most 0.30 together).  The
// scale PME_FIX = PmeDev::fix is a power of two chosen per handle (mdx_pme_setup) from the largest |q sqrt(k_e)| of the system so that
// EIGHT atoms of that charge could sit at full weight on one mesh point without the 32-bit sum wrapping - atoms an Angstrom apart put
// one or two there - and never above 2^25: TIP3P water (15.2 for its oxygen) gets 2^24, i.e. 3e-9 e per count, rounding 1.6e-9 e per
// contribution (an fp32 value near 0.25 e rounds to 1.5e-8 e), headroom +-7 e of net spline-weighted charge per point; OPC's M site
// (24.7) gets 2^23, an ion of charge 3 2^22.  (Round 4 used a fixed 2^25 and said "+-64 e": it forgot the sqrt(k_e), the true headroom
// was +-3.5 e and nothing adapted to large charges.)
constexpr float PME_FIX_MAX = 33554432.0f;          // 2^25
constexpr int PME_TB = 14;    // LDS block edge in mesh points
// k into [0, K): an index a few points outside the mesh (a canvas may be wider than a small mesh: a loop, but no integer division)
__device__ __forceinline__ int pme_wrap(int k, int K) { while (k >= K) k -= K; while (k < 0) k += K; return k; }
__global__ __launch_bounds__(256) void pme_spread_tile_kernel(uint32_t T, const float4* __restrict__ posq,
                                                              const uint8_t* __restrict__ slot_flags, PmeDev g,
                                                              float* __restrict__ Q, const uint32_t* gate, uint32_t thr, uint32_t need,
                                                              const float2* __restrict__ lj, int sel) {
    if (gate && *gate > thr) return;
    __shared__ int s_q[PME_TB * PME_TB * PME_TB];
    __shared__ int s_org[3];
    const uint32_t t = blockIdx.x;
    if (t >= T) return;
    const int tid = threadIdx.x;
    for (int k = tid; k < PME_TB * PME_TB * PME_TB; k += 256) s_q[k] = 0;
    // atom a = tid >> 2 of the tile; its quarter q4 = tid & 3 of the 16 (y, z) offset pairs
    const int atom = tid >> 2, q4 = tid & 3;
    const uint32_t slot = t * MDX_TILE + atom;
    const float4 p = posq[slot];
    const bool live = ((slot_flags[slot] & need) == need) && p.w != 0.f && (!sel || ((__float_as_int(lj[slot].y) < 0) == (sel == 2)));
    int k0[3] = {0, 0, 0}; float w[3] = {0.f, 0.f, 0.f};
    {   // unwrapped mesh coordinates: floor may be < 0 or >= K for an atom just outside the box
        const float x[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float u = (x[a] - g.lo[a]) * g.inv_len[a] * (float)g.K[a];
            const float fl = floorf(u);
            w[a] = u - fl; k0[a] = (int)fl - 3;
        }
    }
    // block origin: minimum k0 over the live atoms of the tile (wave minimum, then the four waves through LDS)
    __shared__ int s_min[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int m = live ? k0[a] : 0x3fffffff;
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) m = min(m, __shfl_xor(m, sft));
        if ((tid & 63) == 0) s_min[tid >> 6][a] = m;
    }
    __syncthreads();
    if (tid < 3) s_org[tid] = min(min(s_min[0][tid], s_min[1][tid]), min(s_min[2][tid], s_min[3][tid]));
    __syncthreads();
    const int ox = s_org[0], oy = s_org[1], oz = s_org[2];
    if (live) {
        float mx[4], my[4], mz[4], dd[4];
        bspline4(w[0], mx, dd); bspline4(w[1], my, dd); bspline4(w[2], mz, dd);
        const int lx = k0[0] - ox, ly = k0[1] - oy, lz = k0[2] - oz;
        const bool fits = lx + 4 <= PME_TB && ly + 4 <= PME_TB && lz + 4 <= PME_TB;
        // this lane: y offset b = q4, all four z offsets, all four x offsets
        const int b = q4;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = p.w * mx[a] * my[b] * mz[c];
                if (fits) {
                    atomicAdd(&s_q[((lx + a) * PME_TB + (ly + b)) * PME_TB + (lz + c)], __float2int_rn(v * g.fix));
                } else {
                    const int kx = pme_wrap(k0[0] + a, g.K[0]), ky = pme_wrap(k0[1] + b, g.K[1]), kz = pme_wrap(k0[2] + c, g.K[2]);
                    atomicAdd(Q + ((size_t)kx * g.K[1] + ky) * g.K[2] + kz, v);
                }
            }
        }
    }
    __syncthreads();
    if (ox == 0x3fffffff) return;      // no live atom in the tile
    for (int k = tid; k < PME_TB * PME_TB * PME_TB; k += 256) {
        const int vi = s_q[k];
        if (vi != 0) {
            const float v = (float)vi * g.unfix;
            const int lz = k % PME_TB, ly = (k / PME_TB) % PME_TB, lx = k / (PME_TB * PME_TB);
            const int kx = pme_wrap(ox + lx, g.K[0]), ky = pme_wrap(oy + ly, g.K[1]), kz = pme_wrap(oz + lz, g.K[2]);
            atomicAdd(Q + ((size_t)kx * g.K[1] + ky) * g.K[2] + kz, v);
        }
    }
}

// Brick spread (round 4; the default of single-GPU handles).  The tile kernel above still ends in ~2200 memory-side f32 atomics
// per tile - 36 M per step at 1 M sites, 0.22-0.29 ms - and needs a cleared mesh (a 32 MB fill per step at 200^3).  Here the
// mesh is cut into bricks of 8..16 points per dimension and the spread becomes four launches without one global f32 atomic:
//   bin      every charged atom computes its mesh cell and spline fractions once and drops a 20-byte record {w, q, cell} into the
//            bucket of the brick that holds its cell (rank inside the bucket: one wave-aggregated atomicAdd per brick and wave);
//   canvas   one workgroup per brick accumulates its bucket (fixed point, ds_add_u32: see PME_FIX) in an LDS canvas of (B + 3)^3
//            points - the brick plus the three planes below it that the order-4 support of its atoms reaches - and STORES it;
//   combine  every mesh point is the sum of the 1..8 canvases that cover it, stored once: no fill, no atomics - and, the sums
//            being integer, a mesh that is bitwise the same from run to run;
//   overflow atoms beyond a bucket's capacity (0.25 atoms per cubic Angstrom; liquid water holds 0.10) are listed by slot and added
//            with global atomics behind the combine pass - in practice an empty launch.
// A first attempt at an atomic-free spread ("block-owned": each workgroup walked every tile that could reach its 16^3 block,
// MDX_PME_SPREAD_BLOCK, removed) spent 317 us at 1 M sites: 40 dependent global loads in a row per workgroup and six atoms read
// for every one that landed.  Buckets make the canvas pass read exactly its own atoms, in independent 64-atom batches.
constexpr int PME_CB_MAX = 19;             // canvas edge: at most 16 + 3 points
constexpr size_t PME_BRICK_MIN_MESH = (size_t)1 << 20;      // mesh points from which the brick spread is the default
struct PmeBrickArgs {
    uint32_t S; const float4* posq; const uint8_t* slot_flags; const float2* lj; PmeDev pg; PmeBrickGeom bg;
    float4* rec; uint32_t* code; uint32_t* slot; uint32_t* count;      // count[nbricks]; [nbricks] = overflow slots of this step; [nbricks + 1] = of all steps; [nbricks + 2] = of this step, kept for the gather
    int keep;                                          // 1: the buckets outlive the spread (pme_gather_brick_kernel empties them)
    uint32_t* ovf; int* scratch; float* Q;
    const uint32_t* gate; uint32_t thr; uint32_t need; int sel;
};
__device__ __forceinline__ int pme_brick_start(int i, int K, int nb) { return (i * K) / nb; }
__device__ __forceinline__ int pme_brick_of(int k, int K, int nb) { return ((k + 1) * nb - 1) / K; }

__global__ __launch_bounds__(256) void pme_bin_kernel(PmeBrickArgs a) {
    if (a.gate && *a.gate > a.thr) return;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < a.S) {
        p = a.posq[s];
        live = ((a.slot_flags[s] & a.need) == a.need) && p.w != 0.f && (!a.sel || ((__float_as_int(a.lj[s].y) < 0) == (a.sel == 2)));
    }
    int k0[3] = {0, 0, 0}; float w[3] = {0.f, 0.f, 0.f};
    uint32_t id = 0xffffffffu, cell = 0u;
    if (live) {
        mesh_coords(p, a.pg, k0, w);                // the very expression the gather uses: k0 = floor(u) - 3
        int bi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int fl = k0[d] + 3;
            const uint32_t ct = a.bg.ctab[a.bg.coff[d] + fl];
            bi[d] = (int)(ct >> 16);
            cell |= (ct & 0xFFFFu) << (8 * d);
        }
        id = (uint32_t)((bi[0] * a.bg.nb[1] + bi[1]) * a.bg.nb[2] + bi[2]);
    }
    // rank inside the bucket: the lanes of a wave that share a brick take consecutive places behind ONE atomic
    const int lane = threadIdx.x & 63;
    uint32_t rank = 0u;
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lid = (uint32_t)__shfl((int)id, leader);
        const unsigned long long same = __ballot(live && id == lid);
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(a.count + lid, (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader);
        if (live && id == lid) rank = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (!live) return;
    if (rank < a.bg.cap) {
        const size_t at = (size_t)id * a.bg.cap + rank;
        a.rec[at] = make_float4(w[0], w[1], w[2], p.w);
        a.code[at] = cell;
        a.slot[at] = s;
    } else {
        a.ovf[atomicAdd(a.count + a.bg.nbricks, 1u)] = s;      // (the list holds S slots: it cannot overflow)
    }
}

__global__ __launch_bounds__(256) void pme_canvas_kernel(PmeBrickArgs a) {
    if (a.gate && *a.gate > a.thr) return;
    __shared__ int s_q[PME_CB_MAX * PME_CB_MAX * PME_CB_MAX];
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const int cb1 = a.bg.cb[1], cb2 = a.bg.cb[2];
    const uint32_t vol = a.bg.stride;
    for (uint32_t k = tid; k < vol; k += 256) s_q[k] = 0;
    __syncthreads();
    const uint32_t n = min(a.count[b], a.bg.cap);
    const float4* rec = a.rec + (size_t)b * a.bg.cap;
    const uint32_t* code = a.code + (size_t)b * a.bg.cap;
    // 64 atoms per pass: atom = tid >> 2, this lane's y offset = tid & 3, 4 x 4 (x, z) points each
    const int yb = tid & 3;
    for (uint32_t i = tid >> 2; i < n; i += 64) {
        const float4 r = rec[i];
        const uint32_t c = code[i];
        const int lx = (int)(c & 255u), ly = (int)((c >> 8) & 255u), lz = (int)((c >> 16) & 255u);
        float mx[4], my[4], mz[4], dd[4];
        bspline4(r.x, mx, dd); bspline4(r.y, my, dd); bspline4(r.z, mz, dd);
        const float qy = r.w * my[yb] * a.pg.fix;
        int* row = s_q + ((size_t)lx * cb1 + (ly + yb)) * cb2 + lz;     // canvas origin = brick start - 3: cell l holds points l .. l + 3
#pragma unroll
        for (int qa = 0; qa < 4; ++qa) {
#pragma unroll
            for (int qc = 0; qc < 4; ++qc) atomicAdd(row + (size_t)qa * cb1 * cb2 + qc, __float2int_rn(qy * mx[qa] * mz[qc]));
        }
    }
    __syncthreads();
    // scratch is ONE padded mesh, the canvases tiled in it brick by brick: a (kx, ky) row of the combine pass then reads one
    // contiguous run per covering (x, y) canvas pair instead of a 64-byte piece per brick
    const int cb0 = a.bg.cb[0];
    const int bz = (int)(b % (uint32_t)a.bg.nb[2]), by = (int)((b / (uint32_t)a.bg.nb[2]) % (uint32_t)a.bg.nb[1]), bx = (int)(b / (uint32_t)(a.bg.nb[2] * a.bg.nb[1]));
    const size_t PY = (size_t)a.bg.nb[1] * cb1, PZ = (size_t)a.bg.nb[2] * cb2;
    int* out = a.scratch + ((size_t)bx * cb0 * PY + (size_t)by * cb1) * PZ + (size_t)bz * cb2;
    {   // (element k = tid, carried forward by 256 per iteration: see pme_gather_brick_kernel)
        int cz = tid % cb2, cy = (tid / cb2) % cb1, cx = tid / (cb2 * cb1);
        const int dz = 256 % cb2, dy = (256 / cb2) % cb1, dx = 256 / (cb2 * cb1);
        for (uint32_t k = tid; k < vol; k += 256) {
            out[((size_t)cx * PY + cy) * PZ + cz] = s_q[k];
            cz += dz; if (cz >= cb2) { cz -= cb2; ++cy; }
            cy += dy; if (cy >= cb1) { cy -= cb1; ++cx; }
            cx += dx;
        }
    }
}

// mesh point (kx, ky, kz): the canvas of its own brick holds it at local + 3; when it lies in the top three planes of its brick
// in a dimension, the next brick's canvas (periodic) holds it too, in its bottom three planes
__global__ __launch_bounds__(256) void pme_combine_kernel(PmeBrickArgs a, size_t n_real) {
    if (a.gate && *a.gate > a.thr) return;
    if (!a.keep && blockIdx.x * 256u + threadIdx.x < a.bg.nbricks) a.count[blockIdx.x * 256u + threadIdx.x] = 0u;      // the buckets are spent: empty for the next step
    // one WAVE per (kx, ky) row of the mesh: the x / y bricks are the same for the whole row (scalar); a lane takes four kz at a
    // time and issues all their loads before it adds (a pass with one load in flight per lane ran at 1.4 TB/s: latency)
    const int K1 = a.pg.K[1], K2 = a.pg.K[2];
    const uint32_t rowi = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (rowi >= (uint32_t)(a.pg.K[0] * K1)) return;
    const int lane = threadIdx.x & 63;
    const int kxy[2] = {(int)(rowi / (unsigned)K1), (int)(rowi % (unsigned)K1)};
    int br[2][2], cc[2][2], nn[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int K = a.pg.K[d], nb = a.bg.nb[d];
        const int b = pme_brick_of(kxy[d], K, nb);
        const int s0 = pme_brick_start(b, K, nb), s1 = pme_brick_start(b + 1, K, nb);
        const int l = kxy[d] - s0;
        br[d][0] = b; cc[d][0] = l + 3; nn[d] = 1; br[d][1] = b; cc[d][1] = 0;
        if (l >= s1 - s0 - 3) { br[d][1] = b + 1 == nb ? 0 : b + 1; cc[d][1] = l - (s1 - s0) + 3; nn[d] = 2; }
    }
    const int nbz = a.bg.nb[2], cb0 = a.bg.cb[0], cb1 = a.bg.cb[1], cb2 = a.bg.cb[2];
    const size_t PY = (size_t)a.bg.nb[1] * cb1, PZ = (size_t)nbz * cb2;
    float* const qrow = a.Q + (size_t)rowi * K2;
    for (int kz0 = 0; kz0 < K2; kz0 += 256) {
        int i0[4], i1[4]; bool ok[4], two[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kz = kz0 + lane + 64 * j;
            ok[j] = kz < K2;
            const int2 zt = a.bg.ztab[ok[j] ? kz : 0];
            two[j] = zt.y >= 0;
            i0[j] = zt.x;
            i1[j] = two[j] ? zt.y : zt.x;
        }
        int v[4] = {0, 0, 0, 0};      // fixed point: the sum is exact and does not depend on which workgroup finished first
        for (int ix = 0; ix < nn[0]; ++ix)
            for (int iy = 0; iy < nn[1]; ++iy) {
                const int* row = a.scratch + ((size_t)(br[0][ix] * cb0 + cc[0][ix]) * PY + (size_t)(br[1][iy] * cb1 + cc[1][iy])) * PZ;
                int p0[4], p1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { p0[j] = row[i0[j]]; p1[j] = row[i1[j]]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += p0[j] + (two[j] ? p1[j] : 0);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ok[j]) qrow[kz0 + lane + 64 * j] = (float)v[j] * a.pg.unfix;
    }
}

// atoms that did not fit their bucket (listed by slot): the per-atom form, with global atomics, behind the combine pass; resets the
// overflow count.  One workgroup; 16 lanes per atom as in pme_spread_kernel.
__global__ __launch_bounds__(256) void pme_overflow_kernel(PmeBrickArgs a) {
    if (a.gate && *a.gate > a.thr) return;
    const uint32_t n_all = a.count[a.bg.nbricks];
    if (threadIdx.x == 0) a.count[a.bg.nbricks + 2] = n_all;
    if (n_all == 0u) return;
    const uint32_t n = n_all;
    const int tid = threadIdx.x, b = (tid >> 2) & 3, c = tid & 3;
    for (uint32_t i = tid >> 4; i < n; i += 16) {
        const float4 p = a.posq[a.ovf[i]];
        int k0[3]; float w[3];
        mesh_coords(p, a.pg, k0, w);
        float mx[4], my[4], mz[4], dd[4];
        bspline4(w[0], mx, dd); bspline4(w[1], my, dd); bspline4(w[2], mz, dd);
        int ky = k0[1] + b; if (ky < 0) ky += a.pg.K[1];
        int kz = k0[2] + c; if (kz < 0) kz += a.pg.K[2];
        const float qbc = p.w * my[b] * mz[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int kx = k0[0] + q; if (kx < 0) kx += a.pg.K[0];
            atomicAdd(a.Q + ((size_t)kx * a.pg.K[1] + ky) * a.pg.K[2] + kz, qbc * mx[q]);
        }
    }
    __syncthreads();
    if (tid == 0) {
        a.count[a.bg.nbricks] = 0u;
        a.count[a.bg.nbricks + 1] += n_all;      // how many atoms ever took this path (diagnostic: mdx_pme_brick_overflows)
    }
}

template <bool ENERGY>
__global__ __launch_bounds__(256) void pme_solve_kernel(size_t n, int K1, int K2, int K3h, int K3, float3 inv_len,
                                                        float pi2_over_beta2, float2* __restrict__ F,
                                                        const float* __restrict__ theta, double* energy,
                                                        const uint32_t* gate, uint32_t thr, double escale) {
    if (gate && *gate > thr) return;
    double e = 0.0, w = 0.0;
    // grid-stride: the energy flavour runs on at most 1024 blocks so that its two f64 atomics per block stay cheap
    // (one pair per wave over 27 k blocks was 2.6 ms of serialised atomics around 20 us of work)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = theta[i];
        float2 f = F[i];
        if (t == 0.f) { F[i] = make_float2(0.f, 0.f); continue; }      // m = 0, and the pad entries of a padded row (never written by the transform)
        if (ENERGY) {
            const int k3 = (int)(i % (size_t)K3h);
            const float mult = (k3 == 0 || (2 * k3 == K3)) ? 1.0f : 2.0f;
            const double ei = 0.5 * (double)(mult * t * (f.x * f.x + f.y * f.y));
            // scalar virial of the reciprocal sum: sum_m E_m (1 - 2 pi^2 m^2 / beta^2)  (= -dE/dlambda
            // under r -> lambda r, L -> lambda L; the B-spline moduli and S(m) do not change)
            const size_t ij = i / (size_t)K3h;
            const int k2 = (int)(ij % (size_t)K2), k1 = (int)(ij / (size_t)K2);
            const float m1 = (float)(k1 <= K1 / 2 ? k1 : k1 - K1) * inv_len.x;
            const float m2 = (float)(k2 <= K2 / 2 ? k2 : k2 - K2) * inv_len.y;
            const float m3 = (float)k3 * inv_len.z;
            e += ei;
            w += ei * (1.0 - 2.0 * (double)(pi2_over_beta2 * (m1 * m1 + m2 * m2 + m3 * m3)));
        }
        f.x *= t; f.y *= t;
        F[i] = f;
    }
    if (ENERGY) {
        __shared__ double s_e[4], s_w[4];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { e += __shfl_xor(e, m); w += __shfl_xor(w, m); }
        if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = e; s_w[threadIdx.x >> 6] = w; }
        __syncthreads();
        if (threadIdx.x == 0) {
            e = s_e[0] + s_e[1] + s_e[2] + s_e[3]; w = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            // (decomposed handle: every rank solves the same mesh; each reports its 1/world share of the energy)
            if (e != 0.0) { atomicAdd(&energy[EN_RECIP], e * escale); atomicAdd(&energy[EN_VIRIAL], w * escale); }
        }
    }
}

// ---- x pass + solve + x pass in ONE kernel (round 5) ---------------------------------------------------------------------------
// hipFFT's 3-D plan at 200^3 is a 100-point R2C pass along z and two strided 200-point passes (y, x) each way; the strided passes run
// at 2.2 TB/s and the x passes bracket the solve: mesh out, mesh in, theta in, mesh out, mesh in - five trips of the 36 MB
// half-complex mesh through memory for what is, per line along x, a transform, a multiplication and the inverse transform.
// Here hipFFT does the batched 2-D transform over (y, z) for every x (81 us forward + inverse at 200^3 against 133 for the 3-D plan,
// tools/ubench/fft_2d_batch.cpp) and this kernel does the rest in one trip: a workgroup takes 16 neighbouring z of one y - K0 rows
// of 128 B - into LDS, runs 16 Stockham transforms of length K0 (radices 4, 2, 3, 5; twiddles from an fp64 table), multiplies by
// theta (energy and virial in the flavour that wants them), transforms back and stores.  y[q + s (r p + u)] = w_n^(p u) sum_t
// x[q + s (p + t m)] w_r^(t u), n -> n / r, s -> s r per pass (autosort: natural order in, natural order out; restated and checked
// against numpy in tests/test_pme_reference.py).  MDX_PME_XPASS=0: the 3-D plan and pme_solve_kernel.
struct XpassArgs {
    float2* F; const float* theta; const float2* tw;
    int K0, K1, K3, pitch, nfac; int fac[16];
    float3 inv_len; float pi2_over_beta2; double* energy; double escale;
    const uint32_t* gate; uint32_t thr;
    float* zero; size_t zero_n; uint32_t tiles;      // workgroups behind the first `tiles` clear the charge mesh for the next step's spread
};
constexpr int XP_TK_LOG = 3, XP_TK = 1 << XP_TK_LOG;      // lines per workgroup (8: 26 KB of LDS at K0 = 200, five workgroups per CU; 16 ran at 56 us, two per CU)
// LDS layout of the tile: row x at x * 9 float2.  (Measured: the kernel is bound by instruction issue - ~4.6 k wave instructions per
// workgroup, a wave64 instruction is four cycles of its SIMD - not by LDS bank conflicts: a conflict-free swizzle of the rows cost
// more integer work than it saved, 44.7 against 39.7 us; 16 lines per workgroup 56 us; load / theta / store folded into the passes 56.6.)
__device__ __forceinline__ int xp_row(int x) { return x * (XP_TK + 1); }
constexpr int XP_ROWS(int n) { return n * (XP_TK + 1); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <bool INV> __device__ __forceinline__ float2 mul_mi(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }   // a * (-i) (forward), a * i (inverse)
// One pass of radix R over the K0 x XP_TK tile in LDS.  The radix is a template parameter and the butterfly loop is unrolled: the
// LDS reads of several butterflies go out together (with the radix a run-time value inside the loop every butterfly waited for its
// own reads: 2 us per pass, 41 us per launch at 200^3).
template <bool INV, int R>
__device__ __forceinline__ void xpass_pass(const float2* __restrict__ cur, float2* __restrict__ oth, const float2* __restrict__ stw,
                                           int N, int n, int s, int kk, int g) {
    const int m = n / R, nb = N / R;
    const float inv_s = 1.0f / (float)s;
#pragma unroll 4
    for (int b = g; b < nb; b += 256 / XP_TK) {
        const int pp = (int)(((float)b + 0.5f) * inv_s), q = b - pp * s;
        const int xi = q + s * pp, di = s * m, xo = q + s * R * pp;      // input rows xi + t di, output rows xo + u s
        const int tws = pp * s;                    // twiddle of output u: w_N^(p u s)
        float2 in[R], o[R];
#pragma unroll
        for (int t = 0; t < R; ++t) in[t] = cur[xp_row(xi + t * di) + kk];
        if (R == 2) {
            o[0] = cadd(in[0], in[1]); o[1] = csub(in[0], in[1]);
        } else if (R == 4) {
            const float2 t0 = cadd(in[0], in[2]), t1 = csub(in[0], in[2]), t2 = cadd(in[1], in[3]), t3 = mul_mi<INV>(csub(in[1], in[3]));
            o[0] = cadd(t0, t2); o[1] = cadd(t1, t3); o[2] = csub(t0, t2); o[3] = csub(t1, t3);
        } else if (R == 3) {
            const float2 t1 = cadd(in[1], in[2]), d = csub(in[1], in[2]);
            const float2 t2 = make_float2(in[0].x - 0.5f * t1.x, in[0].y - 0.5f * t1.y);
            const float2 t3 = mul_mi<INV>(make_float2(0.8660254037844386f * d.x, 0.8660254037844386f * d.y));
            o[0] = cadd(in[0], t1); o[1] = cadd(t2, t3); o[2] = csub(t2, t3);
        } else {      // 5
            const float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f, s1 = 0.9510565162951535f, s2 = 0.5877852522924731f;
            const float2 t1 = cadd(in[1], in[R - 1]), t2 = cadd(in[2], in[R - 2]), t3 = csub(in[1], in[R - 1]), t4 = csub(in[2], in[R - 2]);
            const float2 m1 = make_float2(in[0].x + c1 * t1.x + c2 * t2.x, in[0].y + c1 * t1.y + c2 * t2.y);
            const float2 m2 = make_float2(in[0].x + c2 * t1.x + c1 * t2.x, in[0].y + c2 * t1.y + c1 * t2.y);
            const float2 n1 = mul_mi<INV>(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
            const float2 n2 = mul_mi<INV>(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
            o[0] = make_float2(in[0].x + t1.x + t2.x, in[0].y + t1.y + t2.y);
            o[1] = cadd(m1, n1); o[R - 1] = csub(m1, n1); o[2] = cadd(m2, n2); o[R - 2] = csub(m2, n2);
        }
        oth[xp_row(xo) + kk] = o[0];
#pragma unroll
        for (int u = 1; u < R; ++u) {
            float2 w = stw[tws * u];
            if (INV) w.y = -w.y;
            oth[xp_row(xo + u * s) + kk] = cmul(o[u], w);
        }
    }
}
template <bool INV>
__device__ __forceinline__ void xpass_fft(float2*& cur, float2*& oth, const float2* __restrict__ stw, const XpassArgs& a, int kk, int g) {
    int n = a.K0, s = 1;
    for (int f = 0; f < a.nfac; ++f) {
        const int r = a.fac[f];
        if (r == 4) xpass_pass<INV, 4>(cur, oth, stw, a.K0, n, s, kk, g);
        else if (r == 5) xpass_pass<INV, 5>(cur, oth, stw, a.K0, n, s, kk, g);
        else if (r == 2) xpass_pass<INV, 2>(cur, oth, stw, a.K0, n, s, kk, g);
        else xpass_pass<INV, 3>(cur, oth, stw, a.K0, n, s, kk, g);
        __syncthreads();
        float2* t = cur; cur = oth; oth = t;
        n /= r; s *= r;
    }
}
template <bool ENERGY>
__global__ __launch_bounds__(256) void pme_xpass_solve_kernel(XpassArgs a) {
    if (a.gate && *a.gate > a.thr) return;
    if (blockIdx.x >= a.tiles) {
        // The tile spread (meshes below 2^20 points) adds into a cleared mesh: that was a fill launch per step - 5 of the 101 us of a
        // 23 k-site step.  The forward transform has consumed the charge mesh by now and the inverse one writes the potential to a
        // mesh of its own (PmePlan::phi), so the extra workgroups of this launch clear it.
        const size_t per = (size_t)blockDim.x * 16;
        const size_t i0 = (size_t)(blockIdx.x - a.tiles) * per;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const size_t i = i0 + (size_t)k * blockDim.x + threadIdx.x; if (i < a.zero_n) a.zero[i] = 0.f; }
        return;
    }
    extern __shared__ float2 xp_lds[];
    float2* cur = xp_lds;
    float2* oth = xp_lds + XP_ROWS(a.K0);
    float2* stw = xp_lds + 2 * XP_ROWS(a.K0);
    const int tid = threadIdx.x, kk = tid & (XP_TK - 1), g = tid >> XP_TK_LOG;
    const int ktiles = a.pitch / XP_TK;
    const int y = (int)(blockIdx.x / (unsigned)ktiles), k0 = (int)(blockIdx.x % (unsigned)ktiles) * XP_TK;
    const size_t row = (size_t)a.K1 * a.pitch, base = (size_t)y * a.pitch + k0 + kk;
#pragma unroll 8
    for (int x = g; x < a.K0; x += 256 / XP_TK) cur[xp_row(x) + kk] = a.F[(size_t)x * row + base];
    for (int i = tid; i < a.K0; i += 256) stw[i] = a.tw[i];
    __syncthreads();
    xpass_fft<false>(cur, oth, stw, a, kk, g);
    double e = 0.0, w = 0.0;
    const int k3 = k0 + kk;
#pragma unroll 8
    for (int x = g; x < a.K0; x += 256 / XP_TK) {
        const float t = a.theta[(size_t)x * row + base];      // 0 at m = 0 and in the pad entries of a row
        float2 f = cur[xp_row(x) + kk];
        if (t == 0.f) f = make_float2(0.f, 0.f);
        if (ENERGY && t != 0.f) {
            const float mult = (k3 == 0 || (2 * k3 == a.K3)) ? 1.0f : 2.0f;
            const double ei = 0.5 * (double)(mult * t * (f.x * f.x + f.y * f.y));
            const float m1 = (float)(x <= a.K0 / 2 ? x : x - a.K0) * a.inv_len.x;
            const float m2 = (float)(y <= a.K1 / 2 ? y : y - a.K1) * a.inv_len.y;
            const float m3 = (float)k3 * a.inv_len.z;
            e += ei;
            w += ei * (1.0 - 2.0 * (double)(a.pi2_over_beta2 * (m1 * m1 + m2 * m2 + m3 * m3)));
        }
        f.x *= t; f.y *= t;
        cur[xp_row(x) + kk] = f;
    }
    __syncthreads();
    xpass_fft<true>(cur, oth, stw, a, kk, g);
#pragma unroll 8
    for (int x = g; x < a.K0; x += 256 / XP_TK) a.F[(size_t)x * row + base] = cur[xp_row(x) + kk];
    if (ENERGY) {
        __shared__ double s_e[4], s_w[4];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { e += __shfl_xor(e, m); w += __shfl_xor(w, m); }
        if ((tid & 63) == 0) { s_e[tid >> 6] = e; s_w[tid >> 6] = w; }
        __syncthreads();
        if (tid == 0) {
            e = s_e[0] + s_e[1] + s_e[2] + s_e[3]; w = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            if (e != 0.0) { atomicAdd(&a.energy[EN_RECIP], e * a.escale); atomicAdd(&a.energy[EN_VIRIAL], w * a.escale); }
        }
    }
}

// Alchemical window with the reciprocal sum: environment (F) and coupled molecule (G) are transformed separately; the
// mesh potential is linear in the charges, so E(lambda) = E_FF + E_GG + (1 - lambda) E_x with E_x = sum mult theta Re(F G*)
// (= 2 E_env,mol) and dE/dlambda = -E_x.  Both meshes are multiplied by theta for their inverse transforms.
template <bool ENERGY>
__global__ __launch_bounds__(256) void pme_solve2_kernel(size_t n, int K1, int K2, int K3h, int K3, float3 inv_len,
                                                         float pi2_over_beta2, float2* __restrict__ F, float2* __restrict__ G,
                                                         const float* __restrict__ theta, double* energy, double asc,
                                                         const uint32_t* gate, uint32_t thr, double escale) {
    if (gate && *gate > thr) return;
    double e = 0.0, w = 0.0, ex = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = theta[i];
        float2 f = F[i], g = G[i];
        if (t == 0.f) { F[i] = make_float2(0.f, 0.f); G[i] = make_float2(0.f, 0.f); continue; }      // (as in pme_solve_kernel)
        if (ENERGY) {
            const int k3 = (int)(i % (size_t)K3h);
            const float mult = (k3 == 0 || (2 * k3 == K3)) ? 1.0f : 2.0f;
            const double e_self = 0.5 * (double)(mult * t * (f.x * f.x + f.y * f.y + g.x * g.x + g.y * g.y));
            const double e_x = (double)(mult * t * (f.x * g.x + f.y * g.y));
            const size_t ij = i / (size_t)K3h;
            const int k2 = (int)(ij % (size_t)K2), k1 = (int)(ij / (size_t)K2);
            const float m1 = (float)(k1 <= K1 / 2 ? k1 : k1 - K1) * inv_len.x;
            const float m2 = (float)(k2 <= K2 / 2 ? k2 : k2 - K2) * inv_len.y;
            const float m3 = (float)k3 * inv_len.z;
            const double ei = e_self + asc * e_x;
            e += ei; ex += e_x;
            w += ei * (1.0 - 2.0 * (double)(pi2_over_beta2 * (m1 * m1 + m2 * m2 + m3 * m3)));
        }
        f.x *= t; f.y *= t; g.x *= t; g.y *= t;
        F[i] = f; G[i] = g;
    }
    if (ENERGY) {
        __shared__ double s_e[4], s_w[4], s_x[4];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { e += __shfl_xor(e, m); w += __shfl_xor(w, m); ex += __shfl_xor(ex, m); }
        if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = e; s_w[threadIdx.x >> 6] = w; s_x[threadIdx.x >> 6] = ex; }
        __syncthreads();
        if (threadIdx.x == 0) {
            e = s_e[0] + s_e[1] + s_e[2] + s_e[3]; w = s_w[0] + s_w[1] + s_w[2] + s_w[3]; ex = s_x[0] + s_x[1] + s_x[2] + s_x[3];
            // (decomposed handle: every rank solves the same replicated mesh and the energies are summed over the ranks
            // afterwards - escale = 1 / world; the dU/dlambda word is not part of that sum)
            if (e != 0.0) { atomicAdd(&energy[EN_RECIP], e * escale); atomicAdd(&energy[EN_VIRIAL], w * escale); }
            if (ex != 0.0) atomicAdd(&energy[EN_COUNT + 5], -ex);     // dU/dlambda of the reciprocal sum
        }
    }
}

// phi2 != nullptr (alchemical window): an atom feels its own group's potential plus (1 - lambda) x the other group's
template <bool SEPARATE>
__global__ __launch_bounds__(256) void pme_gather_kernel(uint32_t S, const float4* __restrict__ posq,
                                                         const uint8_t* __restrict__ slot_flags, PmeDev g,
                                                         const float* __restrict__ phi, float4* __restrict__ force,
                                                         const uint32_t* gate, uint32_t thr,
                                                         const float* __restrict__ phi2 = nullptr, const float2* __restrict__ lj = nullptr,
                                                         float asc = 1.f, uint32_t need = 1u) {
    if (gate && *gate > thr) return;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float4 p = posq[s];
    if ((slot_flags[s] & need) != need || p.w == 0.f) {      // need = 3: owned atoms only (slab-decomposed mesh: the canvas holds this rank's block)
        if (SEPARATE) force[s] = make_float4(0.f, 0.f, 0.f, 0.f);   // nothing to add for this slot
        return;
    }
    int k0[3]; float w[3];
    mesh_coords(p, g, k0, w);
    float mx[4], my[4], mz[4], dx[4], dy[4], dz[4];
    bspline4(w[0], mx, dx); bspline4(w[1], my, dy); bspline4(w[2], mz, dz);
    int iz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { int k = k0[2] + c; iz[c] = k < 0 ? k + g.K[2] : k; }
    float fx = 0.f, fy = 0.f, fz = 0.f;
    float ca = 1.f, cb = 0.f;          // weights of phi (environment) and phi2 (coupled molecule)
    if (phi2) { const bool inmol = __float_as_int(lj[s].y) < 0; ca = inmol ? asc : 1.f; cb = inmol ? 1.f : asc; }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int kx = k0[0] + a; if (kx < 0) kx += g.K[0];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int ky = k0[1] + b; if (ky < 0) ky += g.K[1];
            const size_t ro = ((size_t)kx * g.K[1] + ky) * g.K[2];
            const float* row = phi + ro;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = row[iz[c]];
                if (phi2) v = ca * v + cb * phi2[ro + iz[c]];
                s0 += mz[c] * v; s1 += dz[c] * v;
            }
            fx += dx[a] * my[b] * s0;
            fy += mx[a] * dy[b] * s0;
            fz += mx[a] * my[b] * s1;
        }
    }
    // SEPARATE: the reciprocal force goes to its own array (the chain runs on a side stream beside the pair and bonded
    // kernels, which are busy with `force`); the caller adds it once both streams have met again
    float4 f = SEPARATE ? make_float4(0.f, 0.f, 0.f, 0.f) : force[s];
    f.x -= p.w * fx * g.scale[0]; f.y -= p.w * fy * g.scale[1]; f.z -= p.w * fz * g.scale[2];
    force[s] = f;
}

// the atoms that did not fit their bucket (the spread's overflow list, its length kept in count[nbricks + 2]): per-atom form, by the
// extra workgroup behind the bricks' (in practice it finds the list empty)
template <bool SEPARATE>
__device__ __forceinline__ void pme_gather_overflow(const PmeBrickArgs& a, const float* __restrict__ phi, float4* __restrict__ force) {
    const uint32_t n = a.count[a.bg.nbricks + 2];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t s = a.ovf[i];
        const float4 p = a.posq[s];
        int k0[3]; float w[3];
        mesh_coords(p, a.pg, k0, w);
        float mx[4], my[4], mz[4], dx[4], dy[4], dz[4];
        bspline4(w[0], mx, dx); bspline4(w[1], my, dy); bspline4(w[2], mz, dz);
        float fx = 0.f, fy = 0.f, fz = 0.f;
        for (int qa = 0; qa < 4; ++qa) {
            int kx = k0[0] + qa; if (kx < 0) kx += a.pg.K[0];
            for (int qb = 0; qb < 4; ++qb) {
                int ky = k0[1] + qb; if (ky < 0) ky += a.pg.K[1];
                const float* row = phi + ((size_t)kx * a.pg.K[1] + ky) * a.pg.K[2];
                float s0 = 0.f, s1 = 0.f;
                for (int qc = 0; qc < 4; ++qc) { int kz = k0[2] + qc; if (kz < 0) kz += a.pg.K[2]; const float v = row[kz]; s0 += mz[qc] * v; s1 += dz[qc] * v; }
                fx += dx[qa] * my[qb] * s0; fy += mx[qa] * dy[qb] * s0; fz += mx[qa] * my[qb] * s1;
            }
        }
        float4 f = SEPARATE ? make_float4(0.f, 0.f, 0.f, 0.f) : force[s];
        f.x -= p.w * fx * a.pg.scale[0]; f.y -= p.w * fy * a.pg.scale[1]; f.z -= p.w * fz * a.pg.scale[2];
        force[s] = f;
    }
}

// Brick gather (round 5): the interpolation of the mesh potential back onto the atoms, brick by brick through LDS.  The per-slot
// kernel above issues 64 scattered 4-byte loads per charge (1 M wave-level gathers per step at 786 k charges: bound by the address
// unit, 50 us); here a workgroup loads the (B + 3)^3 points its brick's atoms can reach ONCE (the canvas of the spread, read
// instead of written: 1.67 x the mesh in row pieces), and every record of the bucket the spread filled - spline fractions, charge,
// cell, and now the slot - takes its 64 values from LDS.  The buckets are emptied here (the combine pass leaves them: keep = 1).
// Slots without a record (uncharged sites, dummies) are not visited: the separate force array is cleared once per list generation.
template <bool SEPARATE>
__global__ __launch_bounds__(256) void pme_gather_brick_kernel(PmeBrickArgs a, const float* __restrict__ phi, float4* __restrict__ force) {
    if (a.gate && *a.gate > a.thr) return;
    __shared__ float s_phi[PME_CB_MAX * PME_CB_MAX * PME_CB_MAX];
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b == a.bg.nbricks) { pme_gather_overflow<SEPARATE>(a, phi, force); return; }
    const int cb1 = a.bg.cb[1], cb2 = a.bg.cb[2];
    const uint32_t vol = a.bg.stride;
    const int bz = (int)(b % (uint32_t)a.bg.nb[2]), by = (int)((b / (uint32_t)a.bg.nb[2]) % (uint32_t)a.bg.nb[1]), bx = (int)(b / (uint32_t)(a.bg.nb[2] * a.bg.nb[1]));
    const int K0 = a.pg.K[0], K1 = a.pg.K[1], K2 = a.pg.K[2];
    const int ox = pme_brick_start(bx, K0, a.bg.nb[0]) - 3, oy = pme_brick_start(by, K1, a.bg.nb[1]) - 3, oz = pme_brick_start(bz, K2, a.bg.nb[2]) - 3;
    const uint32_t n = min(a.count[b], a.bg.cap);
    if (n) {
        // (cx, cy, cz) of element k = tid, then carried forward by 256 per iteration: three integer divisions per thread instead of
        // four per element (27 elements per thread: the divisions were most of this loop)
        int cz = tid % cb2, cy = (tid / cb2) % cb1, cx = tid / (cb2 * cb1);
        const int dz = 256 % cb2, dy = (256 / cb2) % cb1, dx = 256 / (cb2 * cb1);
        for (uint32_t k = tid; k < vol; k += 256) {
            int kx = ox + cx, ky = oy + cy, kz = oz + cz;      // (a canvas is at most K + 2 wide: one wrap either way)
            kx += kx < 0 ? K0 : (kx >= K0 ? -K0 : 0); ky += ky < 0 ? K1 : (ky >= K1 ? -K1 : 0); kz += kz < 0 ? K2 : (kz >= K2 ? -K2 : 0);
            kx = min(max(kx, 0), K0 - 1); ky = min(max(ky, 0), K1 - 1); kz = min(max(kz, 0), K2 - 1);      // (meshes narrower than a canvas: cells no atom of the brick reads)
            s_phi[k] = phi[((size_t)kx * K1 + ky) * K2 + kz];
            cz += dz; if (cz >= cb2) { cz -= cb2; ++cy; }
            cy += dy; if (cy >= cb1) { cy -= cb1; ++cx; }
            cx += dx;
        }
    }
    __syncthreads();
    const float4* rec = a.rec + (size_t)b * a.bg.cap;
    const uint32_t* code = a.code + (size_t)b * a.bg.cap;
    const uint32_t* slot = a.slot + (size_t)b * a.bg.cap;
    for (uint32_t i = tid; i < n; i += 256) {
        const float4 r = rec[i];
        const uint32_t c = code[i], s = slot[i];
        const int lx = (int)(c & 255u), ly = (int)((c >> 8) & 255u), lz = (int)((c >> 16) & 255u);
        float mx[4], my[4], mz[4], dx[4], dy[4], dz[4];
        bspline4(r.x, mx, dx); bspline4(r.y, my, dy); bspline4(r.z, mz, dz);
        float fx = 0.f, fy = 0.f, fz = 0.f;
#pragma unroll
        for (int qa = 0; qa < 4; ++qa) {
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                const float* row = s_phi + ((size_t)(lx + qa) * cb1 + (ly + qb)) * cb2 + lz;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int qc = 0; qc < 4; ++qc) { const float v = row[qc]; s0 += mz[qc] * v; s1 += dz[qc] * v; }
                fx += dx[qa] * my[qb] * s0;
                fy += mx[qa] * dy[qb] * s0;
                fz += mx[qa] * my[qb] * s1;
            }
        }
        float4 f = SEPARATE ? make_float4(0.f, 0.f, 0.f, 0.f) : force[s];
        f.x -= r.w * fx * a.pg.scale[0]; f.y -= r.w * fy * a.pg.scale[1]; f.z -= r.w * fz * a.pg.scale[2];
        force[s] = f;
    }
    if (tid == 0) a.count[b] = 0u;      // (read into n by every thread in front of the barrier above)
}
__global__ __launch_bounds__(256) void pme_add_force_kernel(uint32_t S, float4* __restrict__ force, const float4* __restrict__ add,
                                                            const uint32_t* gate, uint32_t thr) {
    if (gate && *gate > thr) return;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float4 f = force[s];
    const float4 a = add[s];
    f.x += a.x; f.y += a.y; f.z += a.z;
    force[s] = f;
}

// ---- Slab-decomposed SPME (decomposed handles, round 3) ------------------------------------------------------------------
// The first decomposed form all-reduced a replicated mesh (55 MB per rank and step at 240^3) and ran the whole FFT on every rank.
// Now the mesh is cut into x-slabs, one per rank (K1 / W planes):
//   spread     into the full-size canvas as before, but only a rank's own block is ever touched: its brick widened by the
//              drift margin and the spline support (block(q) is a pure function of the decomposition, every rank knows all);
//   charges    my block, cut by slab owner, travels to the owners in ONE send/recv group; an owner adds what arrives
//              (blocks overlap at their margins) into its slab;
//   FFT        2-D R2C on the slab's planes (batched hipFFT) -> transpose group: rank q receives the rows y in its y-slab from
//              everybody and lays them out with x fastest -> 1-D C2C along x (unit stride);
//   solve      theta(m) on the transposed slab; every rank sums the energy and virial of ITS part (no 1 / W factor any more);
//   back       1-D inverse, transpose group, 2-D C2R: the convolved potential on the slab's planes;
//   potential  the reverse of `charges`: every block's cells go back to the block's rank, which writes them into its canvas;
//   gather     as before, owned atoms only.
// Per rank and step at 240^3 on 8 ranks: 2 x ~9 MB of real mesh (block minus the part that stays) + 2 x 6.1 MB of transposes,
// against 55 MB in and out of an all-reduce; FFT work per rank 1 / W.  Falls back to the replicated mesh when K1 or K2 is not a
// multiple of W, a block would wrap onto itself, or an alchemical window is on (two meshes).
// One launch moves all the groups of a direction: blockIdx.y = the group (x-planes [a_lo, a_lo + a_cnt) of the block of rank q).
// MODE 0: slab[i - i0][j][k] += buf (charges arriving at their slab owner; blocks overlap at their margins: atomic add);
//      1: mesh[i][j][k] = buf (potential arriving at the block's rank, mesh = its canvas, i0 = 0);
//      2: buf = slab[i - i0][j][k] (potential leaving the slab owner);  3: buf = mesh[i][j][k] (charges leaving the canvas)
constexpr int PME_MAX_GROUPS = 48;
struct PmeGroupTab {
    int n;
    int q[PME_MAX_GROUPS], a_lo[PME_MAX_GROUPS], a_cnt[PME_MAX_GROUPS]; uint32_t row0[PME_MAX_GROUPS];
    int b0[32][3], nb[32][3];
};
template <int MODE>
__global__ __launch_bounds__(256) void pme_block_move_kernel(PmeGroupTab t, int K1, int K2, int K3, int i0, float* __restrict__ mesh,
                                                             float4* __restrict__ buf) {
    const int g = blockIdx.y;
    const int q = t.q[g], nby = t.nb[q][1], nbz = t.nb[q][2];
    const size_t n = (size_t)t.a_cnt[g] * nby * nbz;
    float* const b = reinterpret_cast<float*>(buf + t.row0[g]);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % (size_t)nbz), bb = (int)((e / (size_t)nbz) % (size_t)nby), a = t.a_lo[g] + (int)(e / ((size_t)nbz * nby));
        int i = (t.b0[q][0] + a) % K1; if (i < 0) i += K1;
        int j = (t.b0[q][1] + bb) % K2; if (j < 0) j += K2;
        int k = (t.b0[q][2] + c) % K3; if (k < 0) k += K3;
        const size_t at = ((size_t)(i - i0) * K2 + j) * K3 + k;
        if (MODE == 0) atomicAdd(&mesh[at], b[e]);
        else if (MODE == 1) mesh[at] = b[e];
        else b[e] = mesh[at];
    }
}
// The transposes go through staging buffers of W blocks of blk = nx ny K3h complex numbers each, padded to whole float4 rows
// (blk_pad).  SLAB side: slab_cplx [nx][K2][K3h] <-> block q holds the rows y in [q ny, (q+1) ny) as [x][y_loc][k].  LINE side:
// tr [ny][K3h][K1] (x fastest: a strided 1-D hipFFT over [K1][ny][K3h] took 54 us at 200^3 where the unit-stride one takes a
// few) <-> block s holds the planes x in [s nx, (s+1) nx) as [x_loc][y_loc][k].  TO_BUF: mesh -> buffer.
template <bool SLAB_SIDE, bool TO_BUF>
__global__ __launch_bounds__(256) void pme_transpose_kernel(size_t n, int nx, int ny, int K2, int K3h, size_t blk, size_t blk_pad,
                                                            float2* __restrict__ mesh, float2* __restrict__ buf) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    size_t o;
    if (SLAB_SIDE) {
        const int k = (int)(e % (size_t)K3h), y = (int)((e / (size_t)K3h) % (size_t)K2), x = (int)(e / ((size_t)K3h * K2));
        const int q = y / ny, yl = y - q * ny;
        o = (size_t)q * blk_pad + ((size_t)x * ny + yl) * K3h + k;
    } else {      // e runs over tr in ITS order, x fastest: [y_loc][k][x] - the lines along x are contiguous for the 1-D transform
        const int K1 = (int)(n / ((size_t)ny * K3h));
        const int x = (int)(e % (size_t)K1), k = (int)((e / (size_t)K1) % (size_t)K3h), yl = (int)(e / ((size_t)K1 * K3h));
        const int sidx = x / nx, xl = x - sidx * nx;
        o = (size_t)sidx * blk_pad + ((size_t)xl * ny + yl) * K3h + k;
    }
    if (TO_BUF) buf[o] = mesh[e]; else mesh[e] = buf[o];
}
template <bool ENERGY>
__global__ __launch_bounds__(256) void pme_solve_slab_kernel(size_t n, int K1, int K2, int K3h, int K3, int ny, int y0, float3 inv_len,
                                                             float pi2_over_beta2, float2* __restrict__ F,
                                                             const float* __restrict__ theta, double* energy,
                                                             const uint32_t* gate, uint32_t thr) {
    if (gate && *gate > thr) return;
    double e = 0.0, w = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k1 = (int)(i % (size_t)K1), k3 = (int)((i / (size_t)K1) % (size_t)K3h), k2 = y0 + (int)(i / ((size_t)K1 * K3h));      // tr is [y_loc][k][x]
        const float t = theta[((size_t)k1 * K2 + k2) * K3h + k3];
        float2 f = F[i];
        if (ENERGY) {
            const float mult = (k3 == 0 || (2 * k3 == K3)) ? 1.0f : 2.0f;
            const double ei = 0.5 * (double)(mult * t * (f.x * f.x + f.y * f.y));
            const float m1 = (float)(k1 <= K1 / 2 ? k1 : k1 - K1) * inv_len.x;
            const float m2 = (float)(k2 <= K2 / 2 ? k2 : k2 - K2) * inv_len.y;
            const float m3 = (float)k3 * inv_len.z;
            e += ei;
            w += ei * (1.0 - 2.0 * (double)(pi2_over_beta2 * (m1 * m1 + m2 * m2 + m3 * m3)));
        }
        f.x *= t; f.y *= t;
        F[i] = f;
    }
    if (ENERGY) {
        __shared__ double s_e[4], s_w[4];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { e += __shfl_xor(e, m); w += __shfl_xor(w, m); }
        if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = e; s_w[threadIdx.x >> 6] = w; }
        __syncthreads();
        if (threadIdx.x == 0) {
            e = s_e[0] + s_e[1] + s_e[2] + s_e[3]; w = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            if (e != 0.0) { atomicAdd(&energy[EN_RECIP], e); atomicAdd(&energy[EN_VIRIAL], w); }
        }
    }
}

// ---- host -------------------------------------------------------------------------------------------
static int good_size(double min_n) {
    for (int n = std::max(8, (int)std::ceil(min_n));; ++n) {
        int m = n;
        for (int p : {2, 3, 5}) while (m % p == 0) m /= p;
        if (m == 1 && n % 2 == 0) return n;
    }
}

static std::vector<double> bspline_moduli4(int K) {
    const double mn[4] = {0.0, 1.0 / 6.0, 4.0 / 6.0, 1.0 / 6.0};   // M4 at the knots 0..3
    std::vector<double> b2(K);
    for (int m = 0; m < K; ++m) {
        std::complex<double> den(0.0, 0.0);
        for (int k = 0; k < 4; ++k) den += mn[k] * std::exp(std::complex<double>(0.0, 2.0 * M_PI * m * k / K));
        b2[m] = 1.0 / std::max(std::norm(den), 1e-30);
    }
    return b2;
}

static void pme_slab_free(PmePlan* p);
static void pme_brick_free(PmePlan* p) {
    for (void** q : {(void**)&p->brick.ctab, (void**)&p->brick.ztab, (void**)&p->brick.rec, (void**)&p->brick.code, (void**)&p->brick.slot, (void**)&p->brick.count, (void**)&p->brick.ovf, (void**)&p->brick.scratch})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    p->brick.on = false;
}

// Brick geometry of a mesh: ceil(K / edge) bricks per dimension, brick i starting at floor(i K / n) - sizes differ by at most one
// point and lie in [edge / 2, edge] (K >= 8), so the support of an atom (3 points below its cell) never reaches past the
// neighbouring brick.  MDX_PME_SPREAD_BRICK=0 keeps the tile kernel; MDX_PME_BRICK_EDGE = 8..16; MDX_PME_BRICK_CAP forces a
// bucket capacity (tests use it to drive atoms through the overflow list).
static int pme_brick_setup(mdx_handle* h, PmePlan* p) {
    pme_brick_free(p);
    const char* const oe = std::getenv("MDX_PME_SPREAD_BRICK");       // (read at every setup: a test can choose per handle)
    if ((oe && oe[0] == '0') || h->dd || h->n_local != h->N) return MDX_OK;
    const char* ee = std::getenv("MDX_PME_BRICK_EDGE");
    const size_t n_real = p->n_real;
    // small meshes keep the tile kernel (fill + one launch against four launches: 6.9 k against 6.1 k steps/s at 23 k sites, 56^3);
    // MDX_PME_SPREAD_BRICK=1 forces the bricks
    if (!(oe && oe[0] == '1') && n_real < PME_BRICK_MIN_MESH) return MDX_OK;
    int edge = ee ? std::atoi(ee) : (n_real >= (size_t)1 << 21 ? 16 : 8);      // small meshes: more, smaller workgroups
    edge = std::min(16, std::max(8, edge));
    PmeBrickGeom& g = p->brick.g;
    double vol_cell = 1.0;
    for (int d = 0; d < 3; ++d) {
        const int K = h->pme_K[d];
        g.nb[d] = (K + edge - 1) / edge;
        int bmax = 0;
        for (int i = 0; i < g.nb[d]; ++i) bmax = std::max(bmax, ((i + 1) * K) / g.nb[d] - (i * K) / g.nb[d]);
        g.cb[d] = bmax + 3;
        if (g.cb[d] > PME_CB_MAX) FAIL(MDX_EDEVICE, "PME brick larger than its canvas");
        vol_cell *= ((double)h->box_hi[d] - (double)h->box_lo[d]) / K;
    }
    g.nbricks = (uint32_t)(g.nb[0] * g.nb[1] * g.nb[2]);
    g.stride = (uint32_t)(g.cb[0] * g.cb[1] * g.cb[2]);
    // capacity of a bucket: 0.25 charged atoms per cubic Angstrom of the largest brick (liquid water: 0.10), a multiple of 64
    const double bvol = vol_cell * (g.cb[0] - 3) * (g.cb[1] - 3) * (g.cb[2] - 3);
    uint32_t cap = (uint32_t)std::ceil(0.25 * bvol);
    g.cap = (std::min<uint32_t>(std::max<uint32_t>(cap, 1u), h->N + 63u) + 63u) & ~63u;
    if (const char* ce = std::getenv("MDX_PME_BRICK_CAP")) g.cap = (uint32_t)std::max(1, std::atoi(ce));
    HIP_TRY(hipMalloc((void**)&p->brick.rec, sizeof(float4) * (size_t)g.nbricks * g.cap));
    HIP_TRY(hipMalloc((void**)&p->brick.code, sizeof(uint32_t) * (size_t)g.nbricks * g.cap));
    HIP_TRY(hipMalloc((void**)&p->brick.slot, sizeof(uint32_t) * (size_t)g.nbricks * g.cap));
    p->brick.force_zero_epoch = ~0ull;
    HIP_TRY(hipMalloc((void**)&p->brick.count, sizeof(uint32_t) * ((size_t)g.nbricks + 4)));
    HIP_TRY(hipMemset(p->brick.count, 0, sizeof(uint32_t) * ((size_t)g.nbricks + 4)));
    HIP_TRY(hipMalloc((void**)&p->brick.ovf, sizeof(uint32_t) * ((size_t)h->N + 64)));
    HIP_TRY(hipMalloc((void**)&p->brick.scratch, sizeof(int) * (size_t)g.nbricks * g.stride));
    {   // the tables (host mirrors of pme_brick_of / pme_brick_start)
        auto start = [](int i, int K, int nb) { return (i * K) / nb; };
        auto brick_of = [](int k, int K, int nb) { return ((k + 1) * nb - 1) / K; };
        std::vector<uint32_t> ct;
        for (int d = 0; d < 3; ++d) {
            g.coff[d] = (int)ct.size();
            for (int k = 0; k < h->pme_K[d]; ++k) { const int b = brick_of(k, h->pme_K[d], g.nb[d]); ct.push_back(((uint32_t)b << 16) | (uint32_t)(k - start(b, h->pme_K[d], g.nb[d]))); }
        }
        const int K2 = h->pme_K[2], nbz = g.nb[2], cb2 = g.cb[2];
        std::vector<int2> zt(K2);
        for (int kz = 0; kz < K2; ++kz) {
            const int bz = brick_of(kz, K2, nbz), s0 = start(bz, K2, nbz), s1 = start(bz + 1, K2, nbz), l = kz - s0;
            zt[kz].x = bz * cb2 + l + 3;
            zt[kz].y = l >= s1 - s0 - 3 ? (bz + 1 == nbz ? 0 : bz + 1) * cb2 + l - (s1 - s0) + 3 : -1;
        }
        HIP_TRY(hipMalloc((void**)&p->brick.ctab, sizeof(uint32_t) * ct.size()));
        HIP_TRY(hipMalloc((void**)&p->brick.ztab, sizeof(int2) * zt.size()));
        HIP_TRY(hipMemcpy(p->brick.ctab, ct.data(), sizeof(uint32_t) * ct.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->brick.ztab, zt.data(), sizeof(int2) * zt.size(), hipMemcpyHostToDevice));
        g.ctab = p->brick.ctab; g.ztab = p->brick.ztab;
    }
    p->brick.on = true;
    return MDX_OK;
}

// atoms that ever went through the overflow list of the brick spread (a diagnostic; synchronises)
extern "C" int mdx_pme_brick_overflows(mdx_handle* h, uint64_t* n) {
    if (!h || !n) return MDX_EPARAM;
    *n = 0;
    PmePlan* p = (PmePlan*)h->pme_plan;
    if (!p || !p->brick.on) return MDX_OK;
    HIP_TRY(hipDeviceSynchronize());
    uint32_t v = 0;
    HIP_TRY(hipMemcpy(&v, p->brick.count + p->brick.g.nbricks + 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n = v;
    return MDX_OK;
}

// ---------------------------------------------------------------------------------------------
// Compute units of its own for the reciprocal-space chain: MEASURED AND LEFT OFF (MDX_PME_CUS=n turns it on: n CUs per XCD for the
// chain, the handle's stream on the other 32 - n; hipExtStreamCreateWithCUMask, mask bit b = XCD b % 8, CU b / 8 of it - read back
// with HW_ID / XCC_ID by tools/ubench/cu_mask.hip).  The idea: on plain streams the chain does not overlap - the pair kernel is 16 k
// one-wave workgroups that refill every wave slot the moment it frees, a 256-thread workgroup with LDS never finds its slots
// together, and the chain's first kernel waits for the pair kernel's tail (pme_bin_kernel 19 -> 363 us; 44 of 395 us hidden) - while
// with disjoint CU masks the two run side by side without disturbing each other (ubench: a VALU kernel and a copy, each as fast as
// alone on its CUs).  The price decides it: the pair kernel slows by exactly the CUs it gives up (534 -> 710 us per launch on 24 of
// 32) and the chain's kernels are NOT the streaming kind that loses little on fewer CUs (a plain copy keeps 45 % of its bandwidth
// on a quarter of the CUs; the gather takes 3.6x, canvas 3.0x, combine 3.4x, the FFT passes 1.6-2.0x as long):
// 1,048,576 sites, steps/s: no split 773 | 4 CUs per XCD 509 | 6: 512 | 8: 748 | 10: 654  (profiles/r05_cu_mask.txt).
// ---------------------------------------------------------------------------------------------
int mdx_pme_cu_split(const mdx_handle* h, const mdx_config* c) {
    if (c->coulomb_mode != MDX_COULOMB_EWALD || (c->overrides & MDX_OVR_LONG_RANGE_RECIP_DISABLED)) return 0;
    const char* const o = std::getenv("MDX_PME_OVERLAP");
    if (o ? o[0] != '1' : h->N < 262144u) return 0;              // (the same rule as pme_overlap below, and the explicit A/B arms keep the plain streams)
    int n = 0;
    if (const char* e = std::getenv("MDX_PME_CUS")) n = std::atoi(e);
    if (n <= 0 || n >= 24) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) != hipSuccess || prop.multiProcessorCount != 256) return 0;     // 8 XCDs x 32 CUs: the layout the mask is written for
    return n;
}
int mdx_stream_create_masked(hipStream_t* s, int cus_per_xcd, bool complement) {
    uint32_t m[8] = {};
    for (int b = 0; b < 256; ++b)
        if (((b >> 3) < cus_per_xcd) != complement) m[b >> 5] |= 1u << (b & 31);
    HIP_TRY(hipExtStreamCreateWithCUMask(s, 8, m));
    return MDX_OK;
}
int mdx_stream_unmask(mdx_handle* h) {
    if (!h->pme_cus_per_xcd) return MDX_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->stream_pme) {
        HIP_TRY(hipStreamSynchronize(h->stream_pme));
        (void)hipEventDestroy(h->ev_pme_fork); (void)hipEventDestroy(h->ev_pme_join);
        h->ev_pme_fork = nullptr; h->ev_pme_join = nullptr;
        (void)hipStreamDestroy(h->stream_pme);
        h->stream_pme = nullptr;
    }
    (void)hipStreamDestroy(h->stream);
    h->stream = nullptr; h->pme_cus_per_xcd = 0;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    return MDX_OK;
}

void mdx_pme_destroy(mdx_handle* h) {
    if (h->stream_pme) {
        (void)hipStreamSynchronize(h->stream_pme);
        (void)hipEventDestroy(h->ev_pme_fork); (void)hipEventDestroy(h->ev_pme_join);
        h->ev_pme_fork = nullptr; h->ev_pme_join = nullptr;
        (void)hipStreamDestroy(h->stream_pme);
        h->stream_pme = nullptr;
    }
    PmePlan* p = (PmePlan*)h->pme_plan;
    if (!p) return;
    if (p->have_plans) { p->destroy(p->fwd); p->destroy(p->inv); }
    if (p->have_xpass) { p->destroy(p->fwd2); p->destroy(p->inv2); p->have_xpass = false; }
    if (p->tw) { (void)hipFree(p->tw); p->tw = nullptr; }
    if (p->phi) { (void)hipFree(p->phi); p->phi = nullptr; }
    pme_slab_free(p);
    pme_brick_free(p);
    if (p->lib) dlclose(p->lib);
    delete p;
    h->pme_plan = nullptr;
}

static void pme_slab_free(PmePlan* p) {
    PmePlan::Slab& sl = p->slab;
    if (sl.have_plans) { p->destroy(sl.fwd2d); p->destroy(sl.inv2d); p->destroy(sl.fft1d); sl.have_plans = false; }
    for (void** q : {(void**)&sl.slab_real, (void**)&sl.slab_cplx, (void**)&sl.tr, (void**)&sl.buf_a, (void**)&sl.buf_b})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    sl.cap_a = sl.cap_b = 0; sl.on = false;
}

// Geometry, message lists, buffers and plans of the slab-decomposed mesh; called at every mdx_pme_setup (attach, box change).
// MDX_PME_SLAB=0 keeps the replicated mesh (A/B).
static int pme_slab_setup(mdx_handle* h, PmePlan* p) {
    h->pme_canvas_clean = false; h->pme_clear_pending = false;      // (the block this rank clears and touches may be another from here on)
    PmePlan::Slab& sl = p->slab;
    pme_slab_free(p);
    MdxDecomp* dd = h->dd;
    const char* e = std::getenv("MDX_PME_SLAB");
    if (!dd || dd->world < 2 || dd->world > 32 || (e && e[0] == '0') || !p->plan_many || !p->exec_c2c) return MDX_OK;
    const int W = dd->world, K1 = h->pme_K[0], K2 = h->pme_K[1], K3 = h->pme_K[2], K3h = K3 / 2 + 1;
    if (K1 % W || K2 % W) return MDX_OK;
    const int nx = K1 / W, ny = K2 / W;
    sl.W = W; sl.rank = dd->rank; sl.nx = nx; sl.ny = ny;
    const int K[3] = {K1, K2, K3};
    // every rank's spread block: its brick widened by what an owned atom may have drifted since the partition (margin), the
    // reach of a cluster from its anchor (ext) and the spline support (3 cells below, 1 above the atom's cell)
    const double pad = (double)dd->margin + (double)dd->ext + 1.0;
    for (int q = 0; q < W; ++q) {
        const int cq[3] = {q / (dd->grid[1] * dd->grid[2]), (q / dd->grid[2]) % dd->grid[1], q % dd->grid[2]};
        for (int d = 0; d < 3; ++d) {
            const double L = (double)h->box_hi[d] - (double)h->box_lo[d], hc = L / K[d];
            if (dd->grid[d] == 1) { sl.b0[q][d] = 0; sl.nb[q][d] = K[d]; continue; }
            const double blo = L * cq[d] / dd->grid[d], bhi = L * (cq[d] + 1) / dd->grid[d];
            const int lo = (int)std::floor((blo - pad) / hc) - 3, end = (int)std::floor((bhi + pad) / hc) + 2;
            sl.b0[q][d] = lo; sl.nb[q][d] = end - lo;
            if (sl.nb[q][d] >= K[d] - (d == 0 ? nx : 0)) { sl.b0[q][d] = 0; sl.nb[q][d] = K[d]; }   // (a block that nearly wraps: take the whole dimension)
        }
    }
    // cut every block by slab owner: runs of consecutive x-planes that belong to the same slab
    auto groups_of = [&](int q) {
        std::vector<PmePlan::Group> g;
        for (int a = 0; a < sl.nb[q][0]; ++a) {
            int i = (sl.b0[q][0] + a) % K1; if (i < 0) i += K1;
            const int owner = i / nx;
            if (!g.empty() && g.back().peer == owner && g.back().a_lo + g.back().a_cnt == a) g.back().a_cnt++;
            else g.push_back({owner, a, 1, 0u, 0u});
        }
        for (auto& x : g) x.nrows = (uint32_t)(((size_t)x.a_cnt * sl.nb[q][1] * sl.nb[q][2] + 3) / 4);
        return g;
    };
    sl.out_groups = groups_of(sl.rank);
    uint32_t r = 0;
    for (auto& g : sl.out_groups) { g.row0 = r; r += g.nrows; sl.s_out.push_back({g.peer, g.row0, g.nrows}); }
    const size_t rows_out = r;
    r = 0;
    for (int q = 0; q < W; ++q)
        for (auto g : groups_of(q))
            if (g.peer == sl.rank) { g.peer = q; g.row0 = r; r += g.nrows; sl.in_groups.push_back(g); sl.r_in.push_back({q, g.row0, g.nrows}); }
    const size_t rows_in = r;
    if (sl.out_groups.size() > (size_t)PME_MAX_GROUPS || sl.in_groups.size() > (size_t)PME_MAX_GROUPS) { pme_slab_free(p); return MDX_OK; }
    sl.max_out = sl.max_in = 0;
    for (const auto& g : sl.out_groups) sl.max_out = std::max(sl.max_out, (size_t)g.a_cnt * sl.nb[sl.rank][1] * sl.nb[sl.rank][2]);
    for (const auto& g : sl.in_groups) sl.max_in = std::max(sl.max_in, (size_t)g.a_cnt * sl.nb[g.peer][1] * sl.nb[g.peer][2]);
    const size_t blk_rows = ((size_t)nx * ny * K3h + 1) / 2, tr_rows = blk_rows * W;      // (blocks padded to whole float4 rows)
    for (int q = 0; q < W; ++q) { sl.s_tr.push_back({q, (uint32_t)(q * blk_rows), (uint32_t)blk_rows}); sl.r_tr.push_back({q, (uint32_t)(q * blk_rows), (uint32_t)blk_rows}); }
    sl.cap_a = std::max(rows_out, tr_rows) + 4; sl.cap_b = std::max(rows_in, tr_rows) + 4;
    HIP_TRY(hipMalloc((void**)&sl.buf_a, sizeof(float4) * sl.cap_a)); HIP_TRY(hipMalloc((void**)&sl.buf_b, sizeof(float4) * sl.cap_b));
    HIP_TRY(hipMalloc((void**)&sl.slab_real, sizeof(float) * (size_t)nx * K2 * K3));
    HIP_TRY(hipMalloc((void**)&sl.slab_cplx, sizeof(float2) * (size_t)nx * K2 * K3h));
    HIP_TRY(hipMalloc((void**)&sl.tr, sizeof(float2) * (size_t)K1 * ny * K3h));
    int n2[2] = {K2, K3};
    int n1[1] = {K1};
    int emb1[1] = {K1};
    if (p->plan_many(&sl.fwd2d, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_R2C, nx) != HIPFFT_SUCCESS ||
        p->plan_many(&sl.inv2d, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2R, nx) != HIPFFT_SUCCESS ||
        p->plan_many(&sl.fft1d, 1, n1, emb1, 1, K1, emb1, 1, K1, HIPFFT_C2C, ny * K3h) != HIPFFT_SUCCESS)
        FAIL(MDX_EDEVICE, "hipfftPlanMany failed (slab-decomposed SPME)");
    sl.have_plans = true;
    p->set_stream(sl.fwd2d, h->stream); p->set_stream(sl.inv2d, h->stream); p->set_stream(sl.fft1d, h->stream);
    sl.on = true;
    return MDX_OK;
}

// The reciprocal-space chain on the slab-decomposed mesh (the canvas h->d.pme_q holds this rank's spread charges on entry and
// the convolved potential over this rank's block on return).
static int pme_slab_chain(mdx_handle* h, PmePlan* p, bool energy, const uint32_t* d_gate, uint32_t thr) {
    PmePlan::Slab& sl = p->slab;
    hipStream_t st = h->stream;
    const int K1 = h->pme_K[0], K2 = h->pme_K[1], K3 = h->pme_K[2], K3h = K3 / 2 + 1, nx = sl.nx, ny = sl.ny, me = sl.rank;
    auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    float* canvas = h->d.pme_q;
    auto table = [&](const std::vector<PmePlan::Group>& gs, bool mine) {
        PmeGroupTab t{};
        t.n = (int)gs.size();
        for (int k = 0; k < t.n; ++k) { t.q[k] = mine ? me : gs[k].peer; t.a_lo[k] = gs[k].a_lo; t.a_cnt[k] = gs[k].a_cnt; t.row0[k] = gs[k].row0; }
        for (int q = 0; q < sl.W; ++q) for (int d = 0; d < 3; ++d) { t.b0[q][d] = sl.b0[q][d]; t.nb[q][d] = sl.nb[q][d]; }
        return t;
    };
    const PmeGroupTab t_out = table(sl.out_groups, true), t_in = table(sl.in_groups, false);
    auto grid2 = [](size_t n_max, int n_groups) { return dim3((unsigned)std::min<size_t>((n_max + 255) / 256, 4096), (unsigned)n_groups); };
    // charges: my block -> the slab owners
    hipLaunchKernelGGL(pme_block_move_kernel<3>, grid2(sl.max_out, t_out.n), dim3(256), 0, st, t_out, K1, K2, K3, 0, canvas, sl.buf_a);
    MDX_TRY(mdx_dd_exchange(h, sl.buf_a, sl.s_out, sl.buf_b, sl.r_in, st));
    HIP_TRY(hipMemsetAsync(sl.slab_real, 0, sizeof(float) * (size_t)nx * K2 * K3, st));
    hipLaunchKernelGGL(pme_block_move_kernel<0>, grid2(sl.max_in, t_in.n), dim3(256), 0, st, t_in, K1, K2, K3, me * nx, sl.slab_real, sl.buf_b);
    // forward transform: planes, transpose, lines
    if (p->exec_r2c(sl.fwd2d, sl.slab_real, (hipfftComplex*)sl.slab_cplx) != HIPFFT_SUCCESS) FAIL(MDX_EDEVICE, "hipfftExecR2C failed (slab)");
    const size_t n_sl = (size_t)nx * K2 * K3h, blk = (size_t)nx * ny * K3h, blk_pad = (blk + 1) & ~(size_t)1;
    hipLaunchKernelGGL((pme_transpose_kernel<true, true>), blocks(n_sl), dim3(256), 0, st, n_sl, nx, ny, K2, K3h, blk, blk_pad, sl.slab_cplx, (float2*)sl.buf_a);
    MDX_TRY(mdx_dd_exchange(h, sl.buf_a, sl.s_tr, sl.buf_b, sl.r_tr, st));
    hipLaunchKernelGGL((pme_transpose_kernel<false, false>), blocks(n_sl), dim3(256), 0, st, n_sl, nx, ny, K2, K3h, blk, blk_pad, sl.tr, (float2*)sl.buf_b);
    if (p->exec_c2c(sl.fft1d, (hipfftComplex*)sl.tr, (hipfftComplex*)sl.tr, HIPFFT_FORWARD) != HIPFFT_SUCCESS) FAIL(MDX_EDEVICE, "hipfftExecC2C failed (slab)");
    // theta(m), energy and virial of this rank's part of reciprocal space
    const size_t n_tr = (size_t)K1 * ny * K3h;
    const dim3 gs((unsigned)std::min<size_t>((n_tr + 255) / 256, energy ? 1024 : (size_t)1 << 30));
    const float3 inv_len = make_float3(p->dev.inv_len[0], p->dev.inv_len[1], p->dev.inv_len[2]);
    const float pb = (float)(M_PI * M_PI / ((double)h->cfg.ewald_alpha * h->cfg.ewald_alpha));
    if (energy) hipLaunchKernelGGL(pme_solve_slab_kernel<true>, gs, dim3(256), 0, st, n_tr, K1, K2, K3h, K3, ny, me * ny, inv_len, pb, sl.tr, h->d.pme_theta,
                                   h->d.energy, d_gate, thr);
    else hipLaunchKernelGGL(pme_solve_slab_kernel<false>, gs, dim3(256), 0, st, n_tr, K1, K2, K3h, K3, ny, me * ny, inv_len, pb, sl.tr, h->d.pme_theta,
                            h->d.energy, d_gate, thr);
    // back: lines, transpose, planes
    if (p->exec_c2c(sl.fft1d, (hipfftComplex*)sl.tr, (hipfftComplex*)sl.tr, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) FAIL(MDX_EDEVICE, "hipfftExecC2C failed (slab)");
    hipLaunchKernelGGL((pme_transpose_kernel<false, true>), blocks(n_sl), dim3(256), 0, st, n_sl, nx, ny, K2, K3h, blk, blk_pad, sl.tr, (float2*)sl.buf_a);
    MDX_TRY(mdx_dd_exchange(h, sl.buf_a, sl.s_tr, sl.buf_b, sl.r_tr, st));
    hipLaunchKernelGGL((pme_transpose_kernel<true, false>), blocks(n_sl), dim3(256), 0, st, n_sl, nx, ny, K2, K3h, blk, blk_pad, sl.slab_cplx, (float2*)sl.buf_b);
    if (p->exec_c2r(sl.inv2d, (hipfftComplex*)sl.slab_cplx, sl.slab_real) != HIPFFT_SUCCESS) FAIL(MDX_EDEVICE, "hipfftExecC2R failed (slab)");
    // potential: every block's cells back to the block's rank
    hipLaunchKernelGGL(pme_block_move_kernel<2>, grid2(sl.max_in, t_in.n), dim3(256), 0, st, t_in, K1, K2, K3, me * nx, sl.slab_real, sl.buf_b);
    MDX_TRY(mdx_dd_exchange(h, sl.buf_b, sl.r_in, sl.buf_a, sl.s_out, st));
    hipLaunchKernelGGL(pme_block_move_kernel<1>, grid2(sl.max_out, t_out.n), dim3(256), 0, st, t_out, K1, K2, K3, 0, canvas, sl.buf_a);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

extern "C" int mdx_pme_info(const mdx_handle* h, int* slab_on, uint64_t* mesh_bytes_sent, uint64_t* transpose_bytes_sent, uint64_t* replicated_mesh_bytes) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    const PmePlan* p = (const PmePlan*)h->pme_plan;
    if (!h->pme_on || !p) FAIL(MDX_EPARAM, "the handle has no reciprocal-space sum");
    const PmePlan::Slab& sl = p->slab;
    uint64_t mesh = 0, tr = 0;
    if (sl.on) {
        for (const auto& g : sl.out_groups) if (g.peer != sl.rank) mesh += (uint64_t)g.nrows * 16u;      // charges out
        for (const auto& g : sl.in_groups) if (g.peer != sl.rank) mesh += (uint64_t)g.nrows * 16u;       // potential back
        tr = 2ull * (uint64_t)(sl.W - 1) * (((uint64_t)sl.nx * sl.ny * (h->pme_K[2] / 2 + 1) + 1) / 2 * 16ull);
    }
    if (slab_on) *slab_on = sl.on ? 1 : 0;
    if (mesh_bytes_sent) *mesh_bytes_sent = mesh;
    if (transpose_bytes_sent) *transpose_bytes_sent = tr;
    if (replicated_mesh_bytes) *replicated_mesh_bytes = (uint64_t)p->n_real * 4ull;
    return MDX_OK;
}

int mdx_pme_setup(mdx_handle* h) {
    const mdx_config& c = h->cfg;
    h->pme_on = c.coulomb_mode == MDX_COULOMB_EWALD && !(c.overrides & MDX_OVR_LONG_RANGE_RECIP_DISABLED) &&
                !(c.overrides & MDX_OVR_COULOMB_DISABLED);
    if (!h->pme_on) return MDX_OK;
    if (!(h->per[0] && h->per[1] && h->per[2])) FAIL(MDX_EPARAM, "SPME needs a fully periodic box");
    if (c.pme_order != 0 && c.pme_order != 4) FAIL(MDX_EPARAM, "only B-spline order 4 is implemented");
    if (h->n_local != h->N && !h->dd) FAIL(MDX_EPARAM, "SPME needs the library's own decomposition (mdx_comm_init) when the box is split");
    PmePlan* p = (PmePlan*)h->pme_plan;
    if (!p) {
        p = new PmePlan();
        h->pme_plan = p;
        // the FFT library is opened only when the reciprocal sum is requested (cf. src/util.rs:1094-1100)
        for (const char* name : {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so.0"}) {
            p->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (p->lib) break;
        }
        if (!p->lib) FAIL(MDX_EDEVICE, "hipFFT not found (libhipfft.so.0): the SPME reciprocal sum needs it");
        p->plan3d = (decltype(p->plan3d))dlsym(p->lib, "hipfftPlan3d");
        p->exec_r2c = (decltype(p->exec_r2c))dlsym(p->lib, "hipfftExecR2C");
        p->exec_c2r = (decltype(p->exec_c2r))dlsym(p->lib, "hipfftExecC2R");
        p->set_stream = (decltype(p->set_stream))dlsym(p->lib, "hipfftSetStream");
        p->destroy = (decltype(p->destroy))dlsym(p->lib, "hipfftDestroy");
        p->plan_many = (decltype(p->plan_many))dlsym(p->lib, "hipfftPlanMany");
        p->exec_c2c = (decltype(p->exec_c2c))dlsym(p->lib, "hipfftExecC2C");
        if (!p->plan3d || !p->exec_r2c || !p->exec_c2r || !p->set_stream || !p->destroy)
            FAIL(MDX_EDEVICE, "hipFFT symbols missing");
    }
    int K[3];
    double L[3];
    for (int d = 0; d < 3; ++d) {
        L[d] = (double)h->box_hi[d] - (double)h->box_lo[d];
        K[d] = c.pme_grid[d] ? (int)c.pme_grid[d] : good_size(L[d] / 1.0);
        if (K[d] < 8 || K[d] > 2048) FAIL(MDX_EPARAM, "pme_grid must be in 8..2048");
    }
    const int K3h = K[2] / 2 + 1;
    // Row pitch of the half-complex mesh.  hipfftPlan3d packs it [K0][K1][K2/2 + 1]: rows of 101 complex numbers at 200^3 start on
    // every 8-byte phase of a cache line, and rocFFT's strided passes read and write them in pieces.  Padded to a multiple of 16
    // complex numbers (128 B; hipfftPlanMany's advanced layout) forward + inverse take 133 us instead of 144 at 200^3, 189 instead of
    // 294 at 256^3 (tools/ubench/fft_pitch.cpp, profiles/r05_fft_pitch.txt).  The pad entries carry theta = 0.  Decomposed handles
    // keep the packed rows (their slab transposes and the replicated-mesh all-reduce index with K2/2 + 1).  MDX_PME_PITCH=0: packed.
    static const bool pad_rows = [] { const char* e = std::getenv("MDX_PME_PITCH"); return !(e && e[0] == '0'); }();
    // (hipfftPlanMany takes the distance between transforms as an int: a padded mesh beyond 2^31 - 1 elements - above ~1290^3 -
    // keeps the packed rows of hipfftPlan3d, which has no such argument)
    const int pitch_pad = (K3h + 15) & ~15;
    const bool pad_fits = (size_t)K[0] * K[1] * (size_t)pitch_pad <= (size_t)INT_MAX && (size_t)K[0] * K[1] * (size_t)K[2] <= (size_t)INT_MAX;
    const int pitch = (pad_rows && !h->dd && p->plan_many && pad_fits) ? pitch_pad : K3h;
    const bool regrid = !p->have_plans || K[0] != h->pme_K[0] || K[1] != h->pme_K[1] || K[2] != h->pme_K[2] || pitch != p->pitch;
    p->n_real = (size_t)K[0] * K[1] * K[2];
    p->n_cplx = (size_t)K[0] * K[1] * pitch;
    p->pitch = pitch;
    if (regrid) {
        if (p->have_plans) { p->destroy(p->fwd); p->destroy(p->inv); p->have_plans = false; }
        if (pitch != K3h) {
            int n[3] = {K[0], K[1], K[2]}, re[3] = {K[0], K[1], K[2]}, cx[3] = {K[0], K[1], pitch};
            if (p->plan_many(&p->fwd, 3, n, re, 1, (int)p->n_real, cx, 1, (int)p->n_cplx, HIPFFT_R2C, 1) != HIPFFT_SUCCESS ||
                p->plan_many(&p->inv, 3, n, cx, 1, (int)p->n_cplx, re, 1, (int)p->n_real, HIPFFT_C2R, 1) != HIPFFT_SUCCESS)
                FAIL(MDX_EDEVICE, "hipfftPlanMany failed");
        } else if (p->plan3d(&p->fwd, K[0], K[1], K[2], HIPFFT_R2C) != HIPFFT_SUCCESS ||
                   p->plan3d(&p->inv, K[0], K[1], K[2], HIPFFT_C2R) != HIPFFT_SUCCESS)
            FAIL(MDX_EDEVICE, "hipfftPlan3d failed");
        p->have_plans = true;
        // the hand-written x pass: K0 batches of the 2-D transform + pme_xpass_solve_kernel (padded rows, K0 <= 512: two LDS buffers)
        if (p->have_xpass) { p->destroy(p->fwd2); p->destroy(p->inv2); p->have_xpass = false; }
        if (p->tw) { (void)hipFree(p->tw); p->tw = nullptr; }
    if (p->phi) { (void)hipFree(p->phi); p->phi = nullptr; }
        static const bool xpass_env = [] { const char* e = std::getenv("MDX_PME_XPASS"); return !(e && e[0] == '0'); }();
        if (xpass_env && pitch != K3h && pitch % XP_TK == 0 && K[0] <= 512) {
            int n2[2] = {K[1], K[2]}, re2[2] = {K[1], K[2]}, cx2[2] = {K[1], pitch};
            if (p->plan_many(&p->fwd2, 2, n2, re2, 1, K[1] * K[2], cx2, 1, K[1] * pitch, HIPFFT_R2C, K[0]) == HIPFFT_SUCCESS) {
                if (p->plan_many(&p->inv2, 2, n2, cx2, 1, K[1] * pitch, re2, 1, K[1] * K[2], HIPFFT_C2R, K[0]) == HIPFFT_SUCCESS) p->have_xpass = true;
                else p->destroy(p->fwd2);
            }
            if (p->have_xpass) {
                p->nfac = 0;
                int rest = K[0];
                for (int r : {4, 2, 3, 5})
                    while (rest % r == 0 && !(r == 2 && rest % 4 == 0)) { p->fac[p->nfac++] = r; rest /= r; }
                if (rest != 1) { p->destroy(p->fwd2); p->destroy(p->inv2); p->have_xpass = false; }      // (good_size gives 2-3-5-smooth meshes; a caller's own pme_grid may not)
            }
            if (p->have_xpass) {
                std::vector<float2> tw(K[0]);
                for (int k = 0; k < K[0]; ++k) tw[k] = make_float2((float)std::cos(-2.0 * M_PI * k / K[0]), (float)std::sin(-2.0 * M_PI * k / K[0]));
                HIP_TRY(hipMalloc((void**)&p->tw, sizeof(float2) * K[0]));
                HIP_TRY(hipMemcpy(p->tw, tw.data(), sizeof(float2) * K[0], hipMemcpyHostToDevice));
                const int lds = (int)(sizeof(float2) * ((size_t)2 * XP_ROWS(K[0]) + K[0]));
                if (lds > 64 * 1024) {
                    if (hipFuncSetAttribute((const void*)pme_xpass_solve_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
                        hipFuncSetAttribute((const void*)pme_xpass_solve_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
                        (void)hipGetLastError(); p->destroy(p->fwd2); p->destroy(p->inv2); p->have_xpass = false;
                    }
                }
            }
        }
        for (void** q : {(void**)&h->d.pme_q, (void**)&h->d.pme_f, (void**)&h->d.pme_theta, (void**)&h->d.pme_q2, (void**)&h->d.pme_f2})
            if (*q) { (void)hipFree(*q); *q = nullptr; }
        HIP_TRY(hipMalloc((void**)&h->d.pme_q, sizeof(float) * p->n_real));
        h->pme_canvas_clean = false; h->pme_canvas2_clean = false; h->pme_clear_pending = false;
        HIP_TRY(hipMalloc((void**)&h->d.pme_f, sizeof(float2) * p->n_cplx));
        HIP_TRY(hipMalloc((void**)&h->d.pme_theta, sizeof(float) * p->n_cplx));
        for (int d = 0; d < 3; ++d) h->pme_K[d] = K[d];
    }
    if (regrid || !((PmePlan*)h->pme_plan)->brick.on) MDX_TRY(pme_brick_setup(h, p));     // (bucket capacity follows the box: set_box re-runs this)
    {   // side stream of the reciprocal-space chain (MDX_PME_OVERLAP=0: everything on the handle's stream; a decomposed
        // handle keeps the chain on its own stream: the mesh all-reduce sits inside it)
        // Below ~65 k atoms the chain's kernels are a few microseconds each and the two cross-stream event hops cost
        // more than running it beside the pair kernel hides (23 k sites: 5730 steps/s on one stream, 4920 on two;
        // 131 k sites: 2410 / 2560) - small systems keep everything on the handle's stream.  MDX_PME_OVERLAP=1 / 0 forces.
        const char* const e = std::getenv("MDX_PME_OVERLAP");      // read at every setup: a test can choose per handle
        const int env = e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : -1;
        h->pme_overlap = !h->dd && (env >= 0 ? env >= 1 : h->N >= 65536u);
        if (h->pme_cus_per_xcd && (!h->pme_overlap || env == 2)) MDX_TRY(mdx_stream_unmask(h));      // (decided at create from the same inputs; a joined handle gets here)
        // (mdx_stream_unmask replaces h->stream: every plan is pointed at its stream again right below, the slab plans in pme_slab_setup)
        // MDX_PME_OVERLAP=2 (A/B): the charge spread stays on the handle's stream, in FRONT of the pair kernel, and only the rest
        // of the chain (FFTs, solve, gather) runs beside it: spread and pair kernel both live on the LDS pipeline (ds_add_f32 /
        // the staged j-atoms) and run no faster side by side than one after the other
        h->pme_spread_main = h->pme_overlap && env == 2;
        if (h->pme_overlap && !h->stream_pme) {
            // (MDX_PME_PRIORITY=1: the chain's stream above the handle's - A/B; round 3 measured it 3 % slower)
            static const bool prio = [] { const char* e = std::getenv("MDX_PME_PRIORITY"); return e && e[0] == '1'; }();
            int lo_p = 0, hi_p = 0;
            if (h->pme_cus_per_xcd) MDX_TRY(mdx_stream_create_masked(&h->stream_pme, h->pme_cus_per_xcd, false));
            else if (prio && hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) == hipSuccess) HIP_TRY(hipStreamCreateWithPriority(&h->stream_pme, hipStreamNonBlocking, hi_p));
            else HIP_TRY(hipStreamCreateWithFlags(&h->stream_pme, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_pme_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_pme_join, hipEventDisableTiming));
        }
        hipStream_t fst = h->pme_overlap ? h->stream_pme : h->stream;
        p->set_stream(p->fwd, fst); p->set_stream(p->inv, fst);
        if (p->have_xpass) { p->set_stream(p->fwd2, fst); p->set_stream(p->inv2, fst); }
    }
    for (int d = 0; d < 3; ++d) {
        p->dev.lo[d] = h->box_lo[d]; p->dev.inv_len[d] = (float)(1.0 / L[d]);
        p->dev.scale[d] = (float)(K[d] / L[d]); p->dev.K[d] = K[d];
    }
    {   // fixed-point scale of the charge canvases: eight atoms of the largest charge at full weight must fit 2^31 counts
        const double qmax = std::max(1e-3, h->q_abs_max * std::sqrt((double)h->cfg.coulomb_k));
        double fix = PME_FIX_MAX;
        while (fix * qmax * 8.0 > 2147483648.0 && fix > 1.0) fix *= 0.5;
        p->dev.fix = (float)fix; p->dev.unfix = (float)(1.0 / fix);
    }
    // theta table in fp64 on the host (depends on the box: recomputed by mdx_set_box)
    const double beta = c.ewald_alpha, V = L[0] * L[1] * L[2];
    std::vector<double> b[3] = {bspline_moduli4(K[0]), bspline_moduli4(K[1]), bspline_moduli4(K[2])};
    std::vector<float> th(p->n_cplx, 0.f);
    for (int i = 0; i < K[0]; ++i) {
        const double m1 = (i <= K[0] / 2 ? i : i - K[0]) / L[0];
        for (int j = 0; j < K[1]; ++j) {
            const double m2 = (j <= K[1] / 2 ? j : j - K[1]) / L[1];
            for (int k = 0; k < K3h; ++k) {
                const double m3 = k / L[2];
                const double msq = m1 * m1 + m2 * m2 + m3 * m3;
                double t = 0.0;
                if (msq > 0.0) t = b[0][i] * b[1][j] * b[2][k] * std::exp(-M_PI * M_PI * msq / (beta * beta)) / (M_PI * V * msq);
                th[((size_t)i * K[1] + j) * pitch + k] = (float)t;
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(h->d.pme_theta, th.data(), sizeof(float) * th.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // constants of the Ewald sum (charges: the ones the pair loop uses, i.e. bonded_only atoms carry none)
    h->ewald_self = -(double)c.coulomb_k * beta / std::sqrt(M_PI) * h->sum_q2;
    h->ewald_background = -M_PI * (double)c.coulomb_k * h->total_charge * h->total_charge / (2.0 * V * beta * beta);
    MDX_TRY(pme_slab_setup(h, p));
    return MDX_OK;
}

// With h->pme_overlap the whole chain (spread, FFT, solve, FFT, gather) runs on a SIDE stream, forked before the pair
// kernel and joined after the bonded gather (mdx_pme_fork / mdx_pme_join around them in compute_forces): the pair kernel
// is VALU-bound, this chain is atomics- and bandwidth-bound.  The gather then writes to pme_force, added at the join.
// The charge mesh must be zero when the spread starts.  Clearing it in FRONT of the spread put a 32 MB fill (200^3) at the head
// of the reciprocal-space chain, on the side stream, where it competes with the pair kernel for the whole chip: the trace of the
// reference's default operating point (1 M sites) showed it at ~0.45 ms per step - as long as spread and both FFTs together.  It
// now runs BEHIND the chain that dirtied the mesh, after the event the step joins on: the fill overlaps the step's light tail
// (bonded terms, constraints, the next drift) and the next chain starts on a clean mesh.
// The charge canvas of a handle.  Slab-decomposed mesh: a rank only ever touches its own spread block (its brick widened by drift
// margin and spline support), and x is the slowest dimension of the mesh - the block's x-planes are one or two contiguous runs, so
// only those are cleared (8 x 1 x 1 ranks: an eighth of the 55 MB of a 240^3 mesh per force call; 2 x 2 x 2: a bit over half).
static int pme_clear_canvas(mdx_handle* h, PmePlan* p, float* Q, hipStream_t st) {
    const int K1 = h->pme_K[0];
    const size_t plane = (size_t)h->pme_K[1] * h->pme_K[2];
    if (p->slab.on && !h->alch_on && p->slab.nb[p->slab.rank][0] < K1) {      // (an alchemical window spreads two replicated meshes: all of each)
        const int nbx = p->slab.nb[p->slab.rank][0];
        int x0 = p->slab.b0[p->slab.rank][0] % K1; if (x0 < 0) x0 += K1;
        const int first = std::min(nbx, K1 - x0);
        HIP_TRY(hipMemsetAsync(Q + (size_t)x0 * plane, 0, sizeof(float) * plane * (size_t)first, st));
        if (nbx > first) HIP_TRY(hipMemsetAsync(Q, 0, sizeof(float) * plane * (size_t)(nbx - first), st));
        return MDX_OK;
    }
    HIP_TRY(hipMemsetAsync(Q, 0, sizeof(float) * p->n_real, st));
    return MDX_OK;
}

static int pme_clear_behind(mdx_handle* h, hipStream_t st) {
    if (!h->pme_clear_pending) return MDX_OK;
    h->pme_clear_pending = false;
    if (h->pme_block_spread_used) return MDX_OK;       // the brick spread stores every point: nothing to clear
    if (h->pme_spread_main) return MDX_OK;             // (the spread runs on the handle's stream: it clears in front of itself, alone on the chip)
    PmePlan* p = (PmePlan*)h->pme_plan;
    MDX_TRY(pme_clear_canvas(h, p, h->d.pme_q, st));
    h->pme_canvas_clean = true;
    if (h->alch_on && h->d.pme_q2) { HIP_TRY(hipMemsetAsync(h->d.pme_q2, 0, sizeof(float) * p->n_real, st)); h->pme_canvas2_clean = true; }
    return MDX_OK;
}

int mdx_pme_fork(mdx_handle* h) {
    if (!h->pme_on || !h->pme_overlap) return MDX_OK;
    HIP_TRY(hipEventRecord(h->ev_pme_fork, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->stream_pme, h->ev_pme_fork, 0));
    return MDX_OK;
}
int mdx_pme_join(mdx_handle* h, const uint32_t* d_gate, uint32_t thr) {
    if (!h->pme_on || !h->pme_overlap) return MDX_OK;
    HIP_TRY(hipEventRecord(h->ev_pme_join, h->stream_pme));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_pme_join, 0));
    MDX_TRY(pme_clear_behind(h, h->stream_pme));       // (behind the join event: the step does not wait for it)
    hipLaunchKernelGGL(pme_add_force_kernel, dim3(div_up(h->S, 256)), dim3(256), 0, h->stream, h->S, h->d.force,
                       h->d.pme_force, d_gate, thr);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

int mdx_launch_pme(mdx_handle* h, bool energy, const uint32_t* d_gate, uint32_t thr) {
    if (!h->pme_on) return MDX_OK;
    PmePlan* p = (PmePlan*)h->pme_plan;
    hipStream_t st = h->pme_overlap ? h->stream_pme : h->stream;      // (non-const: MDX_PME_OVERLAP=2 spreads on the handle's stream)
    const int K3h = p->pitch;      // (the solve kernels index rows by the pitch: pad entries have theta = 0)
    const uint32_t need = h->dd ? 3u : 1u;              // decomposed: every rank spreads the charges it OWNS ...
    const double escale = h->dd ? 1.0 / (double)h->dd->world : 1.0;
    const bool alch = h->alch_on;      // two meshes: environment in pme_q / pme_f, coupled molecule in pme_q2 / pme_f2
    if (alch && !h->d.pme_q2) {
        HIP_TRY(hipMalloc((void**)&h->d.pme_q2, sizeof(float) * p->n_real));
        h->pme_canvas2_clean = false;
        HIP_TRY(hipMalloc((void**)&h->d.pme_f2, sizeof(float2) * p->n_cplx));
    }
    const float asc = (float)(1.0 - h->alch_lambda);
    static const bool per_atom_spread = [] { const char* e = std::getenv("MDX_PME_SPREAD_PER_ATOM"); return e && e[0] == '1'; }();
    if (p->slab.on && !alch) {      // decomposed handle: x-slabs of the mesh, one per rank (above)
        float* Q = h->d.pme_q;
        if (!h->pme_canvas_clean) MDX_TRY(pme_clear_canvas(h, p, Q, st));
        h->pme_canvas_clean = false;
        if (per_atom_spread || !h->in_slot_space)
            hipLaunchKernelGGL(pme_spread_kernel, dim3(div_up(h->S * 16u, 256)), dim3(256), 0, st, h->S, h->d.posq,
                               h->d.slot_flags, p->dev, Q, d_gate, thr, need, h->d.lj, 0);
        else
            hipLaunchKernelGGL(pme_spread_tile_kernel, dim3(h->T), dim3(256), 0, st, h->T, h->d.posq, h->d.slot_flags, p->dev,
                               Q, d_gate, thr, need, h->d.lj, 0);
        MDX_TRY(pme_slab_chain(h, p, energy, d_gate, thr));
        hipLaunchKernelGGL(pme_gather_kernel<false>, dim3(div_up(h->S, 256)), dim3(256), 0, st, h->S, h->d.posq, h->d.slot_flags,
                           p->dev, h->d.pme_q, h->d.force, d_gate, thr, nullptr, h->d.lj, 1.f, 3u);
        HIP_TRY(hipGetLastError());
        h->pme_clear_pending = true;
        if (!h->pme_overlap) MDX_TRY(pme_clear_behind(h, st));
        return MDX_OK;
    }
    // brick gather behind a brick spread (not for the two meshes of an alchemical window); MDX_PME_GATHER_BRICK=0: the per-slot kernel
    static const bool gather_brick_env = [] { const char* e = std::getenv("MDX_PME_GATHER_BRICK"); return !(e && e[0] == '0'); }();
    const bool gather_brick = gather_brick_env && p->brick.on && !per_atom_spread && !h->dd && !alch;
    PmeBrickArgs ga{};
    const bool xpass = p->have_xpass && !alch && !h->dd;      // (the alchemical window solves two meshes together: 3-D plans + pme_solve2_kernel)
    for (int grp = 0; grp < (alch ? 2 : 1); ++grp) {
        float* Q = grp ? h->d.pme_q2 : h->d.pme_q;
        const int sel = alch ? grp + 1 : 0;
        bool& clean = grp ? h->pme_canvas2_clean : h->pme_canvas_clean;
        const hipStream_t st_chain = st;
        const bool spread_main = h->pme_overlap && h->pme_spread_main;
        if (spread_main) st = h->stream;
        if (p->brick.on && !per_atom_spread && !h->dd) {
            // brick spread: every mesh point is stored once by the combine pass - no clear, no global f32 atomics
            PmeBrickArgs ba{};
            ba.S = h->S; ba.posq = h->d.posq; ba.slot_flags = h->d.slot_flags; ba.lj = h->d.lj; ba.pg = p->dev; ba.bg = p->brick.g;
            ba.rec = p->brick.rec; ba.code = p->brick.code; ba.slot = p->brick.slot; ba.count = p->brick.count; ba.ovf = p->brick.ovf;
            ba.keep = gather_brick ? 1 : 0; ga = ba;
            ba.scratch = p->brick.scratch; ba.Q = Q; ba.gate = d_gate; ba.thr = thr; ba.need = need; ba.sel = sel;
            hipLaunchKernelGGL(pme_bin_kernel, dim3(div_up(h->S, 256)), dim3(256), 0, st, ba);
            hipLaunchKernelGGL(pme_canvas_kernel, dim3(ba.bg.nbricks), dim3(256), 0, st, ba);
            hipLaunchKernelGGL(pme_combine_kernel, dim3((unsigned)((h->pme_K[0] * h->pme_K[1] + 3) / 4)), dim3(256), 0, st, ba, p->n_real);
            hipLaunchKernelGGL(pme_overflow_kernel, dim3(1), dim3(256), 0, st, ba);
            clean = false; h->pme_block_spread_used = true;
        } else {
        h->pme_block_spread_used = false;
        if (!clean) HIP_TRY(hipMemsetAsync(Q, 0, sizeof(float) * p->n_real, st));
        clean = false;
        if (per_atom_spread || !h->in_slot_space)
            hipLaunchKernelGGL(pme_spread_kernel, dim3(div_up(h->S * 16u, 256)), dim3(256), 0, st, h->S, h->d.posq,
                               h->d.slot_flags, p->dev, Q, d_gate, thr, need, h->d.lj, sel);
        else
            hipLaunchKernelGGL(pme_spread_tile_kernel, dim3(h->T), dim3(256), 0, st, h->T, h->d.posq, h->d.slot_flags, p->dev,
                               Q, d_gate, thr, need, h->d.lj, sel);
        }
        if (spread_main) {      // the rest of the chain leaves for the side stream here
            HIP_TRY(hipEventRecord(h->ev_pme_fork, h->stream));
            HIP_TRY(hipStreamWaitEvent(st_chain, h->ev_pme_fork, 0));
            st = st_chain;
        }
        // decomposed handle: the meshes are summed over the ranks - a replicated mesh, every rank then solves it and
        // interpolates the forces of its own atoms (the all-reduce is ungated: a collective must be entered by every rank alike)
        if (h->dd && h->dd->world > 1) MDX_TRY(mdx_dd_allreduce_f32(h, Q, p->n_real, st));
        if (p->exec_r2c(xpass ? p->fwd2 : p->fwd, Q, (hipfftComplex*)(grp ? h->d.pme_f2 : h->d.pme_f)) != HIPFFT_SUCCESS) FAIL(MDX_EDEVICE, "hipfftExecR2C failed");
    }
    // the tile / per-atom spread adds into a cleared mesh: with the x-pass kernel in the chain the potential gets a mesh of its own and
    // that launch clears the charge mesh (XpassArgs::zero); the brick spread stores every point and needs neither
    const bool phi_own = xpass && !h->pme_block_spread_used;
    if (phi_own && !p->phi) HIP_TRY(hipMalloc((void**)&p->phi, sizeof(float) * p->n_real));
    const dim3 gs((unsigned)std::min<size_t>((p->n_cplx + 255) / 256, energy ? 1024 : (size_t)1 << 30));
    const float3 inv_len = make_float3(p->dev.inv_len[0], p->dev.inv_len[1], p->dev.inv_len[2]);
    const float pb = (float)(M_PI * M_PI / ((double)h->cfg.ewald_alpha * h->cfg.ewald_alpha));
    if (alch) {
        if (energy) hipLaunchKernelGGL(pme_solve2_kernel<true>, gs, dim3(256), 0, st, p->n_cplx, h->pme_K[0], h->pme_K[1], K3h, h->pme_K[2],
                                       inv_len, pb, h->d.pme_f, h->d.pme_f2, h->d.pme_theta, h->d.energy, (double)asc, d_gate, thr, escale);
        else hipLaunchKernelGGL(pme_solve2_kernel<false>, gs, dim3(256), 0, st, p->n_cplx, h->pme_K[0], h->pme_K[1], K3h, h->pme_K[2],
                                inv_len, pb, h->d.pme_f, h->d.pme_f2, h->d.pme_theta, h->d.energy, (double)asc, d_gate, thr, escale);
        if (p->exec_c2r(p->inv, (hipfftComplex*)h->d.pme_f2, h->d.pme_q2) != HIPFFT_SUCCESS) FAIL(MDX_EDEVICE, "hipfftExecC2R failed");
    } else if (xpass) {
        XpassArgs xa{};
        xa.F = h->d.pme_f; xa.theta = h->d.pme_theta; xa.tw = p->tw;
        xa.K0 = h->pme_K[0]; xa.K1 = h->pme_K[1]; xa.K3 = h->pme_K[2]; xa.pitch = p->pitch; xa.nfac = p->nfac;
        for (int i = 0; i < p->nfac; ++i) xa.fac[i] = p->fac[i];
        xa.inv_len = inv_len; xa.pi2_over_beta2 = pb; xa.energy = h->d.energy; xa.escale = escale; xa.gate = d_gate; xa.thr = thr;
        xa.tiles = (uint32_t)(h->pme_K[1] * (p->pitch / XP_TK));
        if (phi_own) { xa.zero = h->d.pme_q; xa.zero_n = p->n_real; }
        const dim3 gx(xa.tiles + (phi_own ? (unsigned)((p->n_real + 4095) / 4096) : 0u));
        const size_t lds = sizeof(float2) * ((size_t)2 * XP_ROWS(xa.K0) + xa.K0);
        if (energy) hipLaunchKernelGGL(pme_xpass_solve_kernel<true>, gx, dim3(256), lds, st, xa);
        else hipLaunchKernelGGL(pme_xpass_solve_kernel<false>, gx, dim3(256), lds, st, xa);
    } else {
        if (energy) hipLaunchKernelGGL(pme_solve_kernel<true>, gs, dim3(256), 0, st, p->n_cplx, h->pme_K[0], h->pme_K[1], K3h,
                                       h->pme_K[2], inv_len, pb, h->d.pme_f, h->d.pme_theta, h->d.energy, d_gate, thr, escale);
        else hipLaunchKernelGGL(pme_solve_kernel<false>, gs, dim3(256), 0, st, p->n_cplx, h->pme_K[0], h->pme_K[1], K3h,
                                h->pme_K[2], inv_len, pb, h->d.pme_f, h->d.pme_theta, h->d.energy, d_gate, thr, escale);
    }
    float* const phi = phi_own ? p->phi : h->d.pme_q;
    if (p->exec_c2r(xpass ? p->inv2 : p->inv, (hipfftComplex*)h->d.pme_f, phi) != HIPFFT_SUCCESS) FAIL(MDX_EDEVICE, "hipfftExecC2R failed");
    const float* phi2 = alch ? h->d.pme_q2 : nullptr;
    if (gather_brick) {
        const uint64_t epoch = h->rebuild_count ^ ((uint64_t)h->S << 40) ^ (h->in_slot_space ? 1ull << 63 : 0ull);
        if (h->pme_overlap && p->brick.force_zero_epoch != epoch) {      // slots without a record keep their zero until the slots are dealt again
            HIP_TRY(hipMemsetAsync(h->d.pme_force, 0, sizeof(float4) * (size_t)h->S, st));
            p->brick.force_zero_epoch = epoch;
        }
        if (h->pme_overlap) {
            hipLaunchKernelGGL(pme_gather_brick_kernel<true>, dim3(ga.bg.nbricks + 1), dim3(256), 0, st, ga, (const float*)h->d.pme_q, h->d.pme_force);
        } else {
            hipLaunchKernelGGL(pme_gather_brick_kernel<false>, dim3(ga.bg.nbricks + 1), dim3(256), 0, st, ga, (const float*)h->d.pme_q, h->d.force);
        }
    } else if (h->pme_overlap)
        hipLaunchKernelGGL(pme_gather_kernel<true>, dim3(div_up(h->S, 256)), dim3(256), 0, st, h->S, h->d.posq, h->d.slot_flags,
                           p->dev, phi, h->d.pme_force, d_gate, thr, phi2, h->d.lj, asc);
    else
        hipLaunchKernelGGL(pme_gather_kernel<false>, dim3(div_up(h->S, 256)), dim3(256), 0, st, h->S, h->d.posq, h->d.slot_flags,
                           p->dev, phi, h->d.force, d_gate, thr, phi2, h->d.lj, asc);
    HIP_TRY(hipGetLastError());
    if (phi_own) { h->pme_canvas_clean = true; h->pme_clear_pending = false; return MDX_OK; }      // (cleared inside the x-pass launch)
    h->pme_clear_pending = true;
    if (!h->pme_overlap) MDX_TRY(pme_clear_behind(h, st));      // (on the handle's own stream: still behind the gather)
    return MDX_OK;
}
