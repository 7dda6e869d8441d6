// mdx_grid.hip — spatial caches: column grid, counting sort, tile/cluster formation, cluster
// bounding boxes and the tile pair list with exclusion masks.  Hand-written HIP for gfx950.
//
// Replaces what `MdState::new` / `md.rebuild_spatial_caches()` do inside the absent `dynamics`
// crate  [ref: /root/reference src/md/mod.rs:689; src/properties/sol_shrinking_box.rs:632].
// Everything here is integer / bounding-box work, HBM- or latency-bound; it runs once per ~25 steps at 300 K.
//
// Pipeline (all on the handle's stream):
//   bin      atom -> (column, z-bin) cell id, wrap into the box, histogram       (1 thread/atom)
//   scan     exclusive prefix over cells; per-column tile counts and prefix
//   scatter  counting-sort scatter (arbitrary order inside a cell) ...
//   cellsort ... made deterministic: each cell's handful of atoms sorted by (z, atom id)
//   assign   one wavefront per tile: 64 z-consecutive atoms of a column, bitonic sub-sort by
//            y | x | z so that each run of 8 lanes (a cluster) is a compact brick
//   gather   slot-space arrays (posq, lj, vel, ref) from caller-order data; dummies parked far away
//   bbox     per-cluster bounding boxes
//   list     per tile: all (j-cluster, image) within r_list of the tile (bounding boxes; candidates looked up through
//            the column's z-bins), with the per-i-cluster mask, exclusion-bearing entries first + their 64-bit
//            per-lane interaction masks; half list: every cluster pair in ONE tile's list (parity of I + J)
//   prune    the pair kernel's lane mapping asks the atoms: cluster pairs without any atom pair inside r_list lose
//            their mask bit (24 % of them), the plain run is compacted in place
//   remap    bonded index lists caller order -> slot order
#include "mdx_internal.h"
#include <atomic>
#include <chrono>
#include <algorithm>
#include <cfloat>
#include <cmath>

#define WAVE_LDS_SYNC()                                      \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

static inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

// ================================================================================================
// exclusive scan of u32 (n elements; callers append a trailing 0 so out[n-1] is the total)
// ================================================================================================
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        uint32_t x = s_wave[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// n_dev (may be null): the element count is still on the device (fused rebuild: the launch covers an upper bound n, the
// elements beyond *n_dev count as zero and are not written)
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_kernel(const uint32_t* __restrict__ in,
                                                                 uint32_t* __restrict__ out,
                                                                 uint32_t* __restrict__ tile_sums, uint32_t n, const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t s_wave[SCAN_THREADS / 64];
    if (n_dev) n = min(n, *n_dev);
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan_256(sum, s_wave, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(uint32_t* __restrict__ sums, uint32_t nb) {
    __shared__ uint32_t s_wave[SCAN_THREADS / 64];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
        uint32_t i = b0 + threadIdx.x;
        uint32_t v = (i < nb) ? sums[i] : 0u, total;
        uint32_t ex = block_exclusive_scan_256(v, s_wave, &total);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
}

__global__ void scan_add_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ sums, uint32_t n, const uint32_t* __restrict__ n_dev) {
    if (n_dev) n = min(n, *n_dev);
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += sums[i / SCAN_TILE];
}

int mdx_exclusive_scan_u32(mdx_handle* h, const uint32_t* in, uint32_t* out, uint32_t n) {
    return mdx_exclusive_scan_u32_ex(h, in, out, n, h->d.scan_tmp);
}

// Arrays of up to 64 k elements (everything a 23 k-atom system scans): ONE workgroup of 1024 threads walks the array
// 8192 elements at a time, carrying the running total - one launch instead of three (tile scan, scan of the tile sums,
// add), which at this size are all launch latency.
constexpr int SCAN1_THREADS = 1024;
constexpr uint32_t SCAN1_WINDOW = 65536u;
// Exclusive scan of n <= 65536 elements by the whole 1024-thread workgroup, starting from `carry_in`; returns the running
// total behind the window (the same value in every thread).  zero_in: the input is cleared behind the read (the cell
// histogram of the fused rebuild is left ready for the next one).
// Wave w owns the contiguous range [w * cw, (w + 1) * cw), cw a multiple of 256, and reads it as 16-byte vectors, lane after
// lane (coalesced: one 1-KB request per instruction), ALL of it before anything else - up to 16 loads in flight, the data
// stays in registers.  The waves' totals meet in LDS once; then every wave scans its registers and writes.  (Round 2's form
// gave thread k the segment [k * ipt, (k + 1) * ipt): 64 cache lines per load instruction, two passes of ~1 us round
// trips - 29 us for the 60 k cells of a 23 k-atom system, 5 us now.)
template <typename CarryFn>
__device__ __forceinline__ uint32_t scan_window_1024(const uint32_t* in, uint32_t* __restrict__ out, uint32_t n, CarryFn carry_of,
                                                     uint32_t* s_wave /* [16] */, uint32_t* zero_in) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = SCAN1_THREADS / 64, MAXIT = 16;                 // NW * MAXIT * 256 = 65536 elements
    const uint32_t cw = ((n + NW * 256u - 1u) / (NW * 256u)) * 256u;
    const uint32_t b = wave * cw, e = min(n, b + cw);
    const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0u;
    uint4 v[MAXIT];
    uint32_t sum = 0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const uint32_t i = b + (uint32_t)it * 256u + (uint32_t)lane * 4u;
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)it * 256u < cw) {
            if (vec && i + 3u < e) x = *reinterpret_cast<const uint4*>(in + i);
            else {
                if (i < e) x.x = in[i];
                if (i + 1u < e) x.y = in[i + 1u];
                if (i + 2u < e) x.z = in[i + 2u];
                if (i + 3u < e) x.w = in[i + 3u];
            }
        }
        v[it] = x;
        sum += x.x + x.y + x.z + x.w;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) s_wave[wave] = sum;
    __syncthreads();
    uint32_t below = 0, local_total = 0;
    for (int w = 0; w < NW; ++w) { const uint32_t x = s_wave[w]; if (w < wave) below += x; local_total += x; }
    const uint32_t carry_in = carry_of(local_total);      // (a chained window waits for its predecessor here: contains a barrier)
    uint32_t carry = carry_in + below;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if ((uint32_t)it * 256u >= cw) break;
        const uint32_t i = b + (uint32_t)it * 256u + (uint32_t)lane * 4u;
        const uint4 x = v[it];
        const uint32_t t = x.x + x.y + x.z + x.w;
        uint32_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        const uint32_t e0 = carry + inc - t, e1 = e0 + x.x, e2 = e1 + x.y, e3 = e2 + x.z;
        if (vec && i + 3u < e) {
            *reinterpret_cast<uint4*>(out + i) = make_uint4(e0, e1, e2, e3);
            if (zero_in) *reinterpret_cast<uint4*>(zero_in + i) = make_uint4(0u, 0u, 0u, 0u);
        } else {
            if (i < e) { out[i] = e0; if (zero_in) zero_in[i] = 0u; }
            if (i + 1u < e) { out[i + 1u] = e1; if (zero_in) zero_in[i + 1u] = 0u; }
            if (i + 2u < e) { out[i + 2u] = e2; if (zero_in) zero_in[i + 2u] = 0u; }
            if (i + 3u < e) { out[i + 3u] = e3; if (zero_in) zero_in[i + 3u] = 0u; }
        }
        carry += __shfl(inc, 63);
    }
    __syncthreads();      // s_wave is free again
    return carry_in + local_total;
}

// Windows of 65536 elements CHAINED over the workgroups of one launch: workgroup w scans window w; it has its window in
// registers and its local total before it needs the running total of the windows in front, which it takes from workgroup
// w - 1 through a word in memory ((generation << 32) | inclusive total: no reset between launches) - a hop of ~1-2 us.
// The grid is at most a few dozen workgroups, all resident at once, so the spin cannot starve its predecessor.
struct ScanChain { unsigned long long* carry; uint32_t gen; };
__device__ __forceinline__ uint32_t chained_carry(const ScanChain& ch, uint32_t w, uint32_t local_total, uint32_t* s_bcast) {
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        if (w > 0) {
            unsigned long long x;
            do { x = __hip_atomic_load(ch.carry + (w - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((uint32_t)(x >> 32) != ch.gen) __builtin_amdgcn_s_sleep(1); }
            while ((uint32_t)(x >> 32) != ch.gen);
            c = (uint32_t)x;
        }
        __hip_atomic_store(ch.carry + w, ((unsigned long long)ch.gen << 32) | (unsigned long long)(c + local_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_bcast = c;
    }
    __syncthreads();
    return *s_bcast;
}
__global__ __launch_bounds__(SCAN1_THREADS) void scan_chained_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n,
                                                                     const uint32_t* __restrict__ n_dev, ScanChain ch) {
    __shared__ uint32_t s_wave[SCAN1_THREADS / 64];
    __shared__ uint32_t s_bcast;
    if (n_dev) n = min(n, *n_dev);
    const uint32_t w = blockIdx.x, w0 = w * SCAN1_WINDOW;
    const uint32_t nw = w0 < n ? min(SCAN1_WINDOW, n - w0) : 0u;       // (a window behind the end still hands the total on)
    (void)scan_window_1024(in + w0, out + w0, nw, [&](uint32_t tot) { return chained_carry(ch, w, tot, &s_bcast); }, s_wave, nullptr);
}

__global__ __launch_bounds__(SCAN1_THREADS) void scan_one_block_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                       uint32_t n, const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t s_wave[SCAN1_THREADS / 64];
    if (n_dev) n = min(n, *n_dev);
    (void)scan_window_1024(in, out, n, [](uint32_t) { return 0u; }, s_wave, nullptr);
}

static int scan_chain_of(mdx_handle* h, ScanChain* ch) {
    if (!h->d.scan_chain) { HIP_TRY(hipMalloc((void**)&h->d.scan_chain, sizeof(unsigned long long) * 64)); HIP_TRY(hipMemsetAsync(h->d.scan_chain, 0, sizeof(unsigned long long) * 64, h->stream)); }
    ch->carry = h->d.scan_chain; ch->gen = ++h->scan_gen;
    return MDX_OK;
}
static int scan_u32_dev_n(mdx_handle* h, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums, const uint32_t* n_dev) {
    if (n == 0) return MDX_OK;
    uint32_t nb = div_up(n, SCAN_TILE);
    if (nb > 1 && n <= SCAN1_WINDOW) {
        hipLaunchKernelGGL(scan_one_block_kernel, dim3(1), dim3(SCAN1_THREADS), 0, h->stream, in, out, n, n_dev);
        HIP_TRY(hipGetLastError());
        return MDX_OK;
    }
    // up to eight windows: one launch of chained windows (a 200 k-atom rank's role offsets: three ~5 us launches -> one of ~8 us);
    // beyond that the chain of hops costs what the three-launch form's bandwidth does
    static const bool chain_ok = [] { const char* e = std::getenv("MDX_SCAN_CHAINED"); return !(e && e[0] == '0'); }();
    if (chain_ok && n > SCAN1_WINDOW && n <= 8u * SCAN1_WINDOW) {
        ScanChain ch; MDX_TRY(scan_chain_of(h, &ch));
        hipLaunchKernelGGL(scan_chained_kernel, dim3(div_up(n, SCAN1_WINDOW)), dim3(SCAN1_THREADS), 0, h->stream, in, out, n, n_dev, ch);
        HIP_TRY(hipGetLastError());
        return MDX_OK;
    }
    hipLaunchKernelGGL(scan_tile_kernel, dim3(nb), dim3(SCAN_THREADS), 0, h->stream, in, out, sums, n, n_dev);
    if (nb > 1) {
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, h->stream, sums, nb);
        hipLaunchKernelGGL(scan_add_kernel, dim3(div_up(n, 256)), dim3(256), 0, h->stream, out, sums, n, n_dev);
    }
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}
int mdx_exclusive_scan_u32_ex(mdx_handle* h, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums) {
    return scan_u32_dev_n(h, in, out, n, sums, nullptr);
}

// ================================================================================================
// binning
// ================================================================================================
__device__ __forceinline__ float wrap1(float x, float lo, float L) {
    float t = x - floorf((x - lo) / L) * L;
    if (t < lo) t += L;
    if (t >= lo + L) t -= L;
    return t;
}

__global__ void rebuild_clear_kernel(uint32_t* __restrict__ slot_of, uint32_t n_slot_of, uint32_t* __restrict__ cell_count,
                                     uint32_t* __restrict__ cell_cursor, uint32_t n_cells1, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slot_of) slot_of[i] = MDX_INVALID;
    if (i < n_cells1) { cell_count[i] = 0u; cell_cursor[i] = 0u; }
    if (i < 4u) flags[i] = 0u;
}

__global__ void bin_atoms_kernel(float4* __restrict__ pos_orig, uint32_t N, GridParams g, const uint8_t* __restrict__ lflag,
                                 uint32_t* __restrict__ cell_of, uint32_t* __restrict__ cell_count,
                                 uint32_t* __restrict__ nonfinite) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    float4 p = pos_orig[o];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {
        atomicOr(nonfinite, 1u);
        p.x = g.lo[0]; p.y = g.lo[1]; p.z = g.lo[2];
    }
    if (g.per[0]) p.x = wrap1(p.x, g.lo[0], g.len[0]);
    if (g.per[1]) p.y = wrap1(p.y, g.lo[1], g.len[1]);
    if (g.per[2]) p.z = wrap1(p.z, g.lo[2], g.len[2]);
    pos_orig[o] = p;
    int cx = min(g.ncx - 1, max(0, mdx_col_of(g, 0, p.x)));
    int cy = min(g.ncy - 1, max(0, mdx_col_of(g, 1, p.y)));
    int zb = min(g.nzb - 1, max(0, (int)((p.z - g.lo[2]) * g.inv_zbin)));
    const int pop = (g.npop > 1 && (lflag[o] & 1u)) ? 1 : 0;      // ghosts get column sets of their own
    uint32_t cell = (uint32_t)(((pop * g.ncx + cx) * g.ncy + cy) * g.nzb + zb);
    cell_of[o] = cell;
    atomicAdd(&cell_count[cell], 1u);
}

__global__ void column_tiles_kernel(const uint32_t* __restrict__ cell_start, uint32_t ncol, int nzb,
                                    uint32_t* __restrict__ col_tiles) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncol) return;
    if (c == ncol) { col_tiles[c] = 0; return; }
    uint32_t cnt = cell_start[(size_t)(c + 1) * nzb] - cell_start[(size_t)c * nzb];
    col_tiles[c] = (cnt + MDX_TILE - 1) / MDX_TILE;
}

__global__ void tile_col_kernel(const uint32_t* __restrict__ tile_start, uint32_t ncol,
                                uint32_t* __restrict__ tile_col) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    for (uint32_t t = tile_start[c]; t < tile_start[c + 1]; ++t) tile_col[t] = c;
}

__global__ void scatter_kernel(const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ cell_start,
                               uint32_t* __restrict__ cell_cursor, uint32_t* __restrict__ sorted_orig, uint32_t N) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    uint32_t c = cell_of[o];
    uint32_t k = atomicAdd(&cell_cursor[c], 1u);
    sorted_orig[cell_start[c] + k] = o;
}

// Order of every cell's members by (z, atom id): removes the arbitrary order the atomic scatter left, so the whole
// build is deterministic.  One thread per ATOM counts the members of its cell that come before it and writes itself to
// that rank - a handful of independent loads per thread.  (One thread per cell running an insertion sort in global
// memory, a chain of dependent loads per comparison, took 81 us at 1 M atoms and 48 us at 23 k.)
__global__ void cell_rank_kernel(uint32_t N, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ cell_start,
                                 const float4* __restrict__ pos_orig, const uint32_t* __restrict__ unsorted,
                                 uint32_t* __restrict__ sorted_orig) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const uint32_t o = unsorted[p];
    const uint32_t c = cell_of[o];
    const uint32_t b = cell_start[c], e = cell_start[c + 1];
    const float z = pos_orig[o].z;
    // (a cell of thousands of members is not a liquid: non-finite coordinates are parked in cell 0 by the binning pass,
    // which also raises the error the host returns after this kernel - keep the scatter order there instead of counting
    // n^2 / 2 pairs, which for a box of NaNs would hold the device for minutes)
    if (e - b > 4096u) { sorted_orig[p] = o; return; }
    uint32_t rank = 0;
    for (uint32_t j = b; j < e; ++j) {
        const uint32_t oj = unsorted[j];
        const float zj = pos_orig[oj].z;
        rank += (zj < z || (zj == z && oj < o)) ? 1u : 0u;
    }
    sorted_orig[b + rank] = o;
}

// ================================================================================================
// tile assignment + in-tile sub-sort (one wavefront per tile)
// ================================================================================================
struct SortItem { float k; float a, b; uint32_t o; };

template <int GROUP>
__device__ __forceinline__ void bitonic_group(float& key, float& p1, float& p2, uint32_t& o, int lane) {
#pragma unroll
    for (int k = 2; k <= GROUP; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            float ok = __shfl_xor(key, j), o1 = __shfl_xor(p1, j), o2 = __shfl_xor(p2, j);
            uint32_t oo = __shfl_xor(o, j);
            bool asc = ((lane & k) == 0) || (k == GROUP);
            bool lower = (lane & j) == 0;
            bool mine_gt = (key > ok) || (key == ok && o > oo);
            bool mine_lt = (key < ok) || (key == ok && o < oo);
            bool take = (lower == asc) ? mine_gt : mine_lt;
            if (take) { key = ok; p1 = o1; p2 = o2; o = oo; }
        }
    }
}

// 64 lanes sorted by (segment, coordinate AXIS, atom): segments are runs of lanes (an element's segment = the first lane of its
// run), so every element stays inside its run.  AXIS < 0: by segment and atom alone.
template <int AXIS>
__device__ __forceinline__ void bitonic64_seg(int& seg, float& x, float& y, float& z, uint32_t& o, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int os = __shfl_xor(seg, j);
            const float ox = __shfl_xor(x, j), oy = __shfl_xor(y, j), oz = __shfl_xor(z, j);
            const uint32_t oo = __shfl_xor(o, j);
            const float key = AXIS == 0 ? x : (AXIS == 1 ? y : (AXIS == 2 ? z : 0.f));
            const float ok = AXIS == 0 ? ox : (AXIS == 1 ? oy : (AXIS == 2 ? oz : 0.f));
            const bool asc = ((lane & k) == 0) || (k == 64);
            const bool lower = (lane & j) == 0;
            const bool mine_gt = seg > os || (seg == os && (key > ok || (key == ok && o > oo)));
            const bool mine_lt = seg < os || (seg == os && (key < ok || (key == ok && o < oo)));
            const bool take = (lower == asc) ? mine_gt : mine_lt;
            if (take) { seg = os; x = ox; y = oy; z = oz; o = oo; }
        }
    }
}

// Clusters by interaction kind (handles with kind_split).  The 64 atoms of a tile are first parted into those with a Lennard-Jones
// well and NO charge, the rest, and padding; each part is then cut like the whole tile is otherwise - by y, then x, then z - but
// at the multiple of 8 lanes nearest the middle of the run, so that the parts may have any size: a run [a, b) that touches c > 1
// clusters is sorted along the level's axis and cut at lane 8 (floor(a / 8) + ceil(c / 2)).  Three levels bring every run inside
// one cluster (8 -> 4 -> 2 -> 1 clusters).  Four-site water: a tile of 16 molecules becomes 2 clusters of oxygens and 6 of
// hydrogens and M sites; the pruning pass then drops the cluster pairs between the two kinds (6 of the 16 site pairs of two such
// waters interact neither way).  Measured at 1,048,576 sites of OPC (rc 10 + skin 2): 9 % fewer cluster pairs - the single-kind
// clusters are larger boxes, so more of them come within range of one another - pair kernel 820 -> 783 us per step, pruning pass
// 290 -> 345 us per rebuild, 712 -> 742 steps/s.  (A pair-kernel body that also left the Lennard-Jones arithmetic out for
// charge-only entries and the Coulomb arithmetic for well-only ones - 9 and 24 of ~47 VALU instructions - bought another 1.3 %
// of the kernel, 742 -> 747, and lost at 23 k sites: dropped.)
__device__ __forceinline__ void kind_tile_order(float& x, float& y, float& z, uint32_t& o, int group, int lane) {
    int seg = group;                    // 0: well and no charge, 1: the other atoms, 2: padding
    bitonic64_seg<-1>(seg, x, y, z, o, lane);
    const int n0 = __popcll(__ballot(seg == 0)), nv = __popcll(__ballot(seg != 2));
    int a = lane < n0 ? 0 : (lane < nv ? n0 : nv), b = lane < n0 ? n0 : (lane < nv ? nv : 64);
    auto cut = [&]() {
        const int fa = a >> 3, c = ((b + 7) >> 3) - fa;
        if (c > 1) { const int m = 8 * (fa + ((c + 1) >> 1)); if (lane < m) b = m; else a = m; }
    };
    seg = a; bitonic64_seg<1>(seg, x, y, z, o, lane); cut();
    seg = a; bitonic64_seg<0>(seg, x, y, z, o, lane); cut();
    seg = a; bitonic64_seg<2>(seg, x, y, z, o, lane);
}

__global__ __launch_bounds__(256) void assign_tiles_kernel(
    uint32_t T, int nzb, const uint32_t* __restrict__ tile_col, const uint32_t* __restrict__ tile_start,
    const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ sorted_orig,
    const float4* __restrict__ pos_orig, const uint32_t* __restrict__ gid, uint32_t* __restrict__ orig_of,
    uint32_t* __restrict__ slot_of, int kind_split, const uint8_t* __restrict__ lflag, const float* __restrict__ o_qs,
    const float2* __restrict__ o_lj) {
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t > T) return;
    if (t == T) {  // the null tile: all dummies
        orig_of[(size_t)t * MDX_TILE + lane] = MDX_INVALID;
        return;
    }
    const uint32_t c = tile_col[t];
    const uint32_t a0 = cell_start[(size_t)c * nzb] + (t - tile_start[c]) * MDX_TILE;
    const uint32_t aend = cell_start[(size_t)(c + 1) * nzb];
    const uint32_t p = a0 + lane;
    const bool valid = p < aend;
    uint32_t o = valid ? sorted_orig[p] : MDX_INVALID;
    float x = FLT_MAX, y = FLT_MAX, z = FLT_MAX;
    if (valid) { float4 q = pos_orig[o]; x = q.x; y = q.y; z = q.z; }
    if (kind_split) {      // the same order rb_assign_kernel gives: slot assignment is a function of the positions, not of the chain that ran
        int group = 2;
        if (valid) {
            const uint32_t g = gid[o];
            group = (!(lflag[o] & 2u) && o_lj[g].y != 0.f && o_qs[g] == 0.f) ? 0 : 1;
        }
        kind_tile_order(x, y, z, o, group, lane);
    } else {
    // halves by y, quarters by x, eighths (clusters) by z
    bitonic_group<64>(y, x, z, o, lane);
    bitonic_group<32>(x, y, z, o, lane);
    bitonic_group<16>(z, x, y, o, lane);
    }
    const uint32_t slot = t * MDX_TILE + lane;
    orig_of[slot] = o;
    if (o != MDX_INVALID) slot_of[gid[o]] = slot;
}

__global__ void gather_slots_kernel(uint32_t S, const uint32_t* __restrict__ orig_of,
                                    const uint32_t* __restrict__ gid, const uint8_t* __restrict__ lflag,
                                    const float4* __restrict__ pos_orig, const float4* __restrict__ vel_orig,
                                    const float* __restrict__ o_qs, const float2* __restrict__ o_lj,
                                    const float* __restrict__ o_invm, float4* __restrict__ posq,
                                    float2* __restrict__ lj, float4* __restrict__ vel, float4* __restrict__ ref,
                                    float4* __restrict__ force, uint8_t* __restrict__ slot_flags, float* __restrict__ path,
                                    float* __restrict__ dprune) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    uint32_t o = orig_of[s];
    float4 p, v; float2 l;
    uint8_t fl = 0;
    if (o != MDX_INVALID) {
        const uint32_t g = gid[o];
        const bool ghost = (lflag[o] & 1u) != 0;
        float4 q = pos_orig[o], w = vel_orig[o];
        // (lflag bit 1: a ghost kept only as the bonded partner of an owned atom - its pairs belong to other ranks)
        const bool silent = (lflag[o] & 2u) != 0;
        p = make_float4(q.x, q.y, q.z, silent ? 0.f : o_qs[g]);
        v = ghost ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(w.x, w.y, w.z, o_invm[g]);
        l = silent ? make_float2(o_lj[g].x, 0.f) : o_lj[g];
        fl = ghost ? 1 : 3;
    } else {
        // dummy: far away, every dummy at its own coordinate so no two coincide
        float d = MDX_DUMMY_BASE + MDX_DUMMY_STEP * (float)(s & 0xFFFFF);
        p = make_float4(d, d + 17.0f * (float)(s >> 20), d, 0.f);
        v = make_float4(0.f, 0.f, 0.f, 0.f);
        l = make_float2(0.f, 0.f);
    }
    posq[s] = p; vel[s] = v; lj[s] = l; ref[s] = make_float4(p.x, p.y, p.z, 0.f);   // .w: path length (dual list)
    path[s] = 0.f; dprune[s] = 0.f;                                                  // (... or, path split, its own array)
    force[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    slot_flags[s] = fl;
}

__global__ void cluster_bbox_kernel(uint32_t NC, const uint32_t* __restrict__ orig_of,
                                    const float4* __restrict__ posq, const uint8_t* __restrict__ slot_flags,
                                    float4* __restrict__ cl_lo, float4* __restrict__ cl_hi,
                                    const float2* __restrict__ lj, uint8_t* __restrict__ cl_kind) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= NC) return;
    float3 lo = make_float3(3.0e38f, 3.0e38f, 3.0e38f), hi = make_float3(-3.0e38f, -3.0e38f, -3.0e38f);
    int n = 0, n_owned = 0;
    uint32_t kind = 0;
    for (int k = 0; k < MDX_CLUSTER; ++k) {
        uint32_t s = c * MDX_CLUSTER + k;
        if (orig_of[s] == MDX_INVALID) continue;
        n_owned += (slot_flags[s] >> 1) & 1;
        float4 p = posq[s];
        kind |= (lj[s].y != 0.f ? 1u : 0u) | (p.w != 0.f ? 2u : 0u);
        lo.x = fminf(lo.x, p.x); lo.y = fminf(lo.y, p.y); lo.z = fminf(lo.z, p.z);
        hi.x = fmaxf(hi.x, p.x); hi.y = fmaxf(hi.y, p.y); hi.z = fmaxf(hi.z, p.z);
        ++n;
    }
    cl_lo[c] = make_float4(lo.x, lo.y, lo.z, (float)n);
    cl_hi[c] = make_float4(hi.x, hi.y, hi.z, (float)n_owned);   // .w: atoms this rank owns (half list)
    cl_kind[c] = (uint8_t)kind;
}

__global__ void unsort_kernel(uint32_t N, const uint32_t* __restrict__ gid, const uint32_t* __restrict__ slot_of,
                              const float4* __restrict__ posq, const float4* __restrict__ vel,
                              float4* __restrict__ pos_orig, float4* __restrict__ vel_orig) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    uint32_t s = slot_of[gid[o]];
    pos_orig[o] = posq[s];
    vel_orig[o] = vel[s];
}

__global__ void gather_orig_kernel(uint32_t N, const uint32_t* __restrict__ gid, const uint32_t* __restrict__ slot_of,
                                   const float4* __restrict__ slot_arr, float4* __restrict__ orig_arr) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    orig_arr[o] = slot_arr[slot_of[gid[o]]];
}

// Role lists into slot space: count per slot, (scan), then copy with the partners re-indexed.
__global__ void role_count_kernel(uint32_t S, const uint32_t* __restrict__ orig_of, const uint32_t* __restrict__ gid,
                                  const uint8_t* __restrict__ lflag, const uint32_t* __restrict__ role_off_o,
                                  uint32_t* __restrict__ cnt) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > S) return;
    uint32_t n = 0;
    if (s < S) {
        const uint32_t o = orig_of[s];
        if (o != MDX_INVALID && !(lflag[o] & 1u)) {   // ghosts carry no bonded work: their owner does it
            const uint32_t g = gid[o];
            n = role_off_o[g + 1] - role_off_o[g];
        }
    }
    cnt[s] = n;
}

__device__ __forceinline__ void role_fill_body(uint32_t s, uint32_t S, const uint32_t* __restrict__ orig_of, const uint32_t* __restrict__ gid,
                                               const uint8_t* __restrict__ lflag, const uint32_t* __restrict__ slot_of,
                                               const uint32_t* __restrict__ role_off_o, const RoleRec* __restrict__ rec_o,
                                               const uint32_t* __restrict__ role_off_s, RoleRec* __restrict__ rec_s,
                                               uint32_t* __restrict__ err) {
    if (s >= S) return;
    const uint32_t o = orig_of[s];
    if (o == MDX_INVALID || (lflag[o] & 1u)) return;
    const uint32_t g = gid[o];
    const uint32_t b = role_off_o[g], n = role_off_o[g + 1] - b, w = role_off_s[s];
    for (uint32_t k = 0; k < n; ++k) {
        RoleRec r = rec_o[b + k];
        const uint32_t kind = r.meta & 0xFu;
        const int np = kind == ROLE_DIHEDRAL ? 3 : (kind == ROLE_ANGLE ? 2 : 1);
        for (int q = 0; q < np; ++q) {
            const uint32_t sp = slot_of[r.p[q]];
            if (sp == MDX_INVALID) { atomicOr(err, 4u); r.p[q] = s; }   // bonded partner outside the halo
            else r.p[q] = sp;
        }
        rec_s[w + k] = r;
    }
}
__global__ void role_fill_kernel(uint32_t S, const uint32_t* __restrict__ orig_of, const uint32_t* __restrict__ gid,
                                 const uint8_t* __restrict__ lflag, const uint32_t* __restrict__ slot_of,
                                 const uint32_t* __restrict__ role_off_o, const RoleRec* __restrict__ rec_o,
                                 const uint32_t* __restrict__ role_off_s, RoleRec* __restrict__ rec_s,
                                 uint32_t* __restrict__ err) {
    role_fill_body(blockIdx.x * blockDim.x + threadIdx.x, S, orig_of, gid, lflag, slot_of, role_off_o, rec_o, role_off_s, rec_s, err);
}

// ================================================================================================
// tile pair list
// ================================================================================================
constexpr int LB_WAVES = 4;
// Statistics (cluster pairs built / kept) are summed into 64 partial counters, a 128-byte line each, by tile number: one 64-bit
// atomic per tile on ONE address is served every ~15 ns even when nobody waits for its result - 16 k tiles = 0.25 ms, which is
// what BOTH list kernels took at 1 M atoms whatever else was done to them.  Totals: pair_sum_kernel -> pair_count[PC_TOTAL ..].
constexpr uint32_t PC_PARTS = 64, PC_STRIDE = 16, PC_TOTAL = PC_PARTS * PC_STRIDE;
__device__ __forceinline__ unsigned long long* pc_slot(unsigned long long* base, uint32_t t, uint32_t which) {
    return base + (size_t)(t & (PC_PARTS - 1u)) * PC_STRIDE + which;
}
constexpr int LB_REGIONS = 64;   // single-pass build: claim regions of the entry / mask arrays (one cursor line each)
#ifndef LB_HASH_SIZE
#define LB_HASH_SIZE 1024
#endif
#ifndef LB_MAXFLAG_SIZE
#define LB_MAXFLAG_SIZE 512
#endif
#ifndef LB_PLAIN_SIZE
#define LB_PLAIN_SIZE 512
#endif
constexpr int LB_HASH = LB_HASH_SIZE;
constexpr int LB_MAXFLAG = LB_MAXFLAG_SIZE;
constexpr int LB_SPC = 4;        // excluded partners per atom whose slots are kept in registers between the two uses
constexpr int LB_CAND = 256;     // candidate j-tiles buffered between the two phases of the neighbourhood search
constexpr int LB_PLAIN = LB_PLAIN_SIZE;   // single-pass build: plain entries of a tile buffered in LDS before its slice of the list is claimed
constexpr int LB_PLAIN_DD = 1536;   // ... on a half-shell decomposed handle, whose ghost columns at the rim of the halo are slivers with tall tiles
                                    // (1024 until round 4: rank 0 of a 2 x 2 x 1 decomposition of the 1 M-atom box - corner columns 2.4 A x 2.4 A, tiles
                                    // 110 A tall - overflowed it at EVERY rebuild and took the unfused chain: 0.85-1.4 ms per rebuild instead of 0.30;
                                    // mdx_stats.rebuild_fallbacks counts such rebuilds, bench.py reports it per rank)
                                    // (32 KB more LDS per workgroup: with it for everybody the 1 M-atom list build went 0.37 -> 0.43 ms)

__device__ __forceinline__ uint32_t hash_u32(uint32_t k) { return ((k * 2654435761u) >> 22) & (uint32_t)(LB_HASH - 1); }  // 10 bits, cut to the table

__device__ __forceinline__ bool hash_insert(uint32_t* tab, uint32_t key) {
    uint32_t hpos = hash_u32(key);
    for (int probe = 0; probe < LB_HASH; ++probe) {
        uint32_t prev = atomicCAS(&tab[hpos], MDX_INVALID, key);
        if (prev == MDX_INVALID || prev == key) return true;
        hpos = (hpos + 1) & (LB_HASH - 1);
    }
    return false;
}

__device__ __forceinline__ bool hash_contains(const uint32_t* tab, uint32_t key) {
    uint32_t hpos = hash_u32(key);
    for (int probe = 0; probe < LB_HASH; ++probe) {
        uint32_t v = tab[hpos];
        if (v == key) return true;
        if (v == MDX_INVALID) return false;
        hpos = (hpos + 1) & (LB_HASH - 1);
    }
    return false;
}

__device__ __forceinline__ float gap(float lo_a, float hi_a, float lo_b, float hi_b) {
    return fmaxf(0.f, fmaxf(lo_a - hi_b, lo_b - hi_a));
}

__device__ __forceinline__ int floor_div(int a, int b) {
    int q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

struct ListArgs {
    uint32_t T;
    GridParams g;
    float r_build;  // slightly inflated list radius for the bounding-box tests
    const uint32_t* tile_col; const uint32_t* tile_start;
    const float4* cl_lo; const float4* cl_hi;
    const uint32_t* orig_of; const uint32_t* slot_of; const uint32_t* gid; const uint8_t* slot_flags;
    const uint32_t* excl_off; const uint32_t* excl_idx;
    // count pass out
    ListCounts* counts; uint32_t* entry_cnt; uint32_t* mchunk_cnt;
    // fill pass in/out (the single-pass build writes them: a tile claims its slice with two atomics)
    uint32_t* entry_off; uint32_t* mchunk_off;
    // single pass: the entry and mask arrays are cut into n_regions equal regions, tile t claims from region t % n_regions with
    // ONE returning 64-bit atomic (low word: entries, high word: masked chunks) on that region's own 128-byte line
    unsigned long long* cursors;
    uint32_t n_regions, region_cap_e, region_cap_m;
    uint2* entries; unsigned long long* masks;
    uint32_t* err;
    uint32_t null_cluster;
    int mask_layout;                 // 1: bit (8e+jj) of lane i-atom; 2: bit (8e+ci) of lane (ii, jj)
    unsigned long long* pair_count;  // statistics: sum of popcount(imask)
    int half;                        // 1: every cluster pair appears in exactly one tile's list
    const uint32_t* cell_start;      // [ncol * nzb + 1]: first sorted atom of every (column, z-bin) cell
    const float4* posq;              // slot-space coordinates, for the exact test of borderline cluster pairs
    // single pass: workgroups list_grid .. of the launch translate the bonded role lists into slot space (role_fill_body)
    uint32_t list_grid, rf_S;
    // fused rebuild: the tile count is still on the device when this kernel is launched (for an upper bound of tiles):
    // rb_ctl[0] = T; null_cluster and rf_S follow from it
    const uint32_t* T_dev;
    const uint32_t* rf_role_off_o; const RoleRec* rf_rec_o; const uint32_t* rf_role_off_s; RoleRec* rf_rec_s;
    const uint8_t* rf_lflag;
};

// MODE 0: count pass (sizes per tile, then a scan); MODE 1: fill pass; MODE 2: SINGLE pass - the tile's entries are
// buffered in LDS while its neighbourhood is searched ONCE, then the wave claims its slice of the entry array and of the
// mask array with two atomics and writes everything out.  Tiles land in completion order, which nothing depends on
// (every consumer addresses a tile's list by entry_off[t] and its counts).  A tile that does not fit the LDS buffers, or
// a list that outgrows the arrays, raises an error bit and the host falls back to count + fill (with larger arrays).
#ifndef MDX_XCD_SWIZZLE
#define MDX_XCD_SWIZZLE 1     // 0: A/B build without the XCD-aware tile order of the list kernels
#endif
enum { LB_COUNT = 0, LB_FILL = 1, LB_SINGLE = 2 };
template <int MODE, int PLAINCAP = LB_PLAIN>
__global__ __launch_bounds__(LB_WAVES * 64) void build_list_kernel(ListArgs a) {
    constexpr bool FILL = MODE != LB_COUNT;        // entries are produced (to memory, or to LDS first)
    constexpr bool SINGLE = MODE == LB_SINGLE;
    __shared__ uint32_t s_hash[LB_WAVES][LB_HASH];
    __shared__ uint32_t s_fl[LB_WAVES][LB_MAXFLAG];
    __shared__ float s_ibb[LB_WAVES][MDX_CL_PER_TILE][6];
    __shared__ uint2 s_plain[SINGLE ? LB_WAVES : 1][SINGLE ? PLAINCAP : 1];
    __shared__ uint8_t s_mimask[SINGLE ? LB_WAVES : 1][SINGLE ? LB_MAXFLAG : 1];
    __shared__ uint32_t s_cand[LB_WAVES][LB_CAND];   // candidate j-tiles of the neighbourhood search: tile | image code << 27
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (SINGLE && a.T_dev) {                       // (fused rebuild: launched for an upper bound of tiles, the count read here)
        const uint32_t T = __builtin_amdgcn_readfirstlane(*a.T_dev);
        a.T = T; a.null_cluster = T * MDX_CL_PER_TILE; a.rf_S = (T + 1u) * MDX_TILE;
    }
    if (SINGLE && blockIdx.x >= a.list_grid) {     // the role lists ride along: independent of the pair list, and this launch leaves CUs idle
        role_fill_body((blockIdx.x - a.list_grid) * (LB_WAVES * 64) + threadIdx.x, a.rf_S, a.orig_of, a.gid, a.rf_lflag, a.slot_of,
                       a.rf_role_off_o, a.rf_rec_o, a.rf_role_off_s, a.rf_rec_s, a.err);
        return;
    }
    const uint32_t per_xcd = (SINGLE ? a.list_grid : gridDim.x) >> 3;   // (a multiple of 8 workgroups) a contiguous eighth of the tiles per XCD
    const uint32_t t = MDX_XCD_SWIZZLE ? ((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * LB_WAVES + wave : blockIdx.x * LB_WAVES + wave;
    if (t >= a.T) return;

    uint32_t* hash = s_hash[wave];
    uint32_t* fl = s_fl[wave];
    const GridParams& g = a.g;
    // a tile that holds no owned atom (halo copies / padding only) is never an i-tile of a full list;
    // in a half list it still owns its share of the pairs with tiles that do hold owned atoms
    const bool tile_owned = __any((a.slot_flags[t * MDX_TILE + lane] & 2u) != 0);
    if (!tile_owned && !a.half) {
        if (MODE != LB_FILL && lane == 0) {
            a.counts[t].n_masked = 0; a.counts[t].n_plain = 0;
            if (SINGLE) { a.entry_off[t] = 0; a.mchunk_off[t] = 0; } else { a.entry_cnt[t] = 0; a.mchunk_cnt[t] = 0; }
        }
        return;
    }

    for (int k = lane; k < LB_HASH; k += 64) hash[k] = MDX_INVALID;
    // i-cluster boxes and the tile box
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t i_owned = 0;
    if (lane < MDX_CL_PER_TILE) {
        float4 l4 = a.cl_lo[t * MDX_CL_PER_TILE + lane], h4 = a.cl_hi[t * MDX_CL_PER_TILE + lane];
        lo[0] = l4.x; lo[1] = l4.y; lo[2] = l4.z; hi[0] = h4.x; hi[1] = h4.y; hi[2] = h4.z;
        for (int d = 0; d < 3; ++d) { s_ibb[wave][lane][d] = lo[d]; s_ibb[wave][lane][3 + d] = hi[d]; }
        if (h4.w > 0.f) i_owned = 1u << lane;
    }
    i_owned = (uint32_t)__ballot(i_owned != 0) & 0xFFu;   // i-clusters that hold an owned atom
#pragma unroll
    for (int m = 1; m < 8; m <<= 1)
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], m));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], m));
        }
    for (int d = 0; d < 3; ++d) { lo[d] = __shfl(lo[d], 0); hi[d] = __shfl(hi[d], 0); }
    WAVE_LDS_SYNC();

    // exclusion-bearing clusters: our own 8 and those holding an excluded partner of any lane
    const uint32_t myslot = t * MDX_TILE + lane;
    const uint32_t myo = a.orig_of[myslot];
    bool ok_ins = true;
    if (lane < MDX_CL_PER_TILE) ok_ins &= hash_insert(hash, t * MDX_CL_PER_TILE + lane);
    uint32_t eb = 0, ee = 0;
    if (myo != MDX_INVALID) { const uint32_t mg = a.gid[myo]; eb = a.excl_off[mg]; ee = a.excl_off[mg + 1]; }
    // the slots of this lane's first LB_SPC excluded partners stay in registers: the mask phase below needs them again for every
    // masked chunk, and re-reading excl_idx -> slot_of there cost two dependent round trips per partner and chunk (a protein
    // tile: ~12 partners x ~5 chunks); unrolled, the loads of this first pass are independent of one another as well
    uint32_t spc[LB_SPC];
#pragma unroll
    for (int q = 0; q < LB_SPC; ++q) spc[q] = (eb + q < ee) ? a.excl_idx[eb + q] : MDX_INVALID;
#pragma unroll
    for (int q = 0; q < LB_SPC; ++q) if (spc[q] != MDX_INVALID) spc[q] = a.slot_of[spc[q]];
#pragma unroll
    for (int q = 0; q < LB_SPC; ++q)
        if (spc[q] != MDX_INVALID) ok_ins &= hash_insert(hash, spc[q] >> 3);   // absent partner: beyond the halo, out of range
    for (uint32_t k = eb + LB_SPC; k < ee; ++k) {
        const uint32_t sp = a.slot_of[a.excl_idx[k]];
        if (sp != MDX_INVALID) ok_ins &= hash_insert(hash, sp >> 3);
    }
    if (!ok_ins) atomicOr(a.err, 1u);
    WAVE_LDS_SYNC();

    const float r = a.r_build, r2 = r * r;
    uint32_t nm = 0, np = 0, npairs = 0;
    uint32_t ebase = 0, nm_pad_total = 0;
    if (MODE == LB_FILL) { ebase = a.entry_off[t]; nm_pad_total = a.counts[t].n_masked; }
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    int ix0, ix1, iy0, iy1, kz0 = 0, kz1 = 0;
    if (isinf(r) || r > 1.0e30f) {
        ix0 = 0; ix1 = g.ncx - 1; iy0 = 0; iy1 = g.ncy - 1;
    } else {
        ix0 = mdx_col_of(g, 0, lo[0] - r);
        ix1 = mdx_col_of(g, 0, hi[0] + r);
        iy0 = mdx_col_of(g, 1, lo[1] - r);
        iy1 = mdx_col_of(g, 1, hi[1] + r);
        if (!g.per[0]) { ix0 = max(ix0, 0); ix1 = min(ix1, g.ncx - 1); }
        else { ix0 = max(ix0, -g.ncx); ix1 = min(ix1, 2 * g.ncx - 1); }
        if (!g.per[1]) { iy0 = max(iy0, 0); iy1 = min(iy1, g.ncy - 1); }
        else { iy0 = max(iy0, -g.ncy); iy1 = min(iy1, 2 * g.ncy - 1); }
        if (g.per[2]) {
            if (lo[2] - r < g.lo[2]) kz0 = -1;
            if (hi[2] + r >= g.lo[2] + g.len[2]) kz1 = 1;
        }
    }

    // The neighbourhood is searched in two phases (round 3; it was one loop nest over columns, images and z-windows whose every
    // iteration began with a chain of dependent loads - tile_start, cell_start, then the cluster boxes - for a window of ~4 tiles,
    // i.e. 32 of 64 lanes, half of which the half list's parity rule then sent idle: 0.37 ms at 1 M atoms, all of it latency).
    // (A) one lane per (column, population, z-image): the lanes look their columns' z-windows up together - one round trip -
    //     and the candidate j-tiles this tile owns (parity rule) are compacted into LDS, image code attached;
    // (B) eight candidate tiles per pass, lane = j-cluster: the bounding-box tests as before, every lane busy.
    const int nxr = ix1 - ix0 + 1, nyr = iy1 - iy0 + 1, nkz = kz1 - kz0 + 1;
    const uint32_t ncombo = (uint32_t)(nxr * nyr * g.npop * nkz);
    uint32_t* const cand = s_cand[wave];
    uint32_t nc = 0;
    auto drain = [&]() {
        WAVE_LDS_SYNC();
        // (the boxes of the next eight candidate tiles are fetched while this pass tests its own)
        uint32_t cd_n = 0; float4 jl_n = make_float4(0.f, 0.f, 0.f, 0.f), jh_n = jl_n;
        if ((uint32_t)(lane >> 3) < nc) {
            cd_n = cand[lane >> 3];
            const uint32_t jn = (cd_n & 0x7FFFFFFu) * MDX_CL_PER_TILE + (uint32_t)(lane & 7);
            jl_n = a.cl_lo[jn]; jh_n = a.cl_hi[jn];
        }
        for (uint32_t base = 0; base < nc; base += 8) {
            const uint32_t k = base + (uint32_t)(lane >> 3);
            bool pass = false, have = k < nc;
            uint32_t imask = 0, jc = 0, code = 13u;
            const uint32_t cd = cd_n;
            float4 jl = jl_n, jh = jh_n;
            if (k + 8 < nc) {
                cd_n = cand[k + 8];
                const uint32_t jn = (cd_n & 0x7FFFFFFu) * MDX_CL_PER_TILE + (uint32_t)(lane & 7);
                jl_n = a.cl_lo[jn]; jh_n = a.cl_hi[jn];
            }
            if (have) {
                const uint32_t Jt = cd & 0x7FFFFFFu;
                code = cd >> 27;
                jc = Jt * MDX_CL_PER_TILE + (uint32_t)(lane & 7);
                const float sx = (float)((int)(code % 3u) - 1) * g.len[0], sy = (float)((int)((code / 3u) % 3u) - 1) * g.len[1],
                            sz = (float)((int)(code / 9u) - 1) * g.len[2];
                jl.x += sx; jh.x += sx; jl.y += sy; jh.y += sy; jl.z += sz; jh.z += sz;
                float dx = gap(jl.x, jh.x, lo[0], hi[0]);
                float dy = gap(jl.y, jh.y, lo[1], hi[1]);
                float dz = gap(jl.z, jh.z, lo[2], hi[2]);
                if (jl.w > 0.f && dx * dx + dy * dy + dz * dz < r2) {
#pragma unroll
                    for (int ci = 0; ci < MDX_CL_PER_TILE; ++ci) {
                        const float* b = s_ibb[wave][ci];
                        float ex = gap(jl.x, jh.x, b[0], b[3]);
                        float ey = gap(jl.y, jh.y, b[1], b[4]);
                        float ez = gap(jl.z, jh.z, b[2], b[5]);
                        if (ex * ex + ey * ey + ez * ez < r2) imask |= 1u << ci;
                    }
                    if (a.half) {
                        // one owner per pair.  Different tiles: by the parity of I + J (balances
                        // the lists).  Same tile: the lower cluster owns (ci, cj); a cluster's
                        // pair with itself is kept and its mask holds the j > i triangle; its
                        // pair with its own periodic image goes to the "positive" image code.
                        if (Jt == t) {
                            const uint32_t cj = jc % MDX_CL_PER_TILE;
                            const uint32_t self = (code >= 13u) ? 1u : 0u;
                            imask &= ((1u << cj) - 1u) | (self << cj);
                        }
                        if (jh.w <= 0.f) imask &= i_owned;   // neither side owned here: not ours
                    }
                    pass = imask != 0;
                }
            }
            const bool flagged = pass && hash_contains(hash, jc);
            const unsigned long long bm = __ballot(flagged), bp = __ballot(pass && !flagged);
            if (FILL) {
                const uint2 ent = make_uint2(jc, code | (imask << 8));
                if (flagged) {
                    uint32_t kk = nm + __popcll(bm & lt_mask);
                    if (!SINGLE) a.entries[ebase + kk] = ent;
                    if (kk < LB_MAXFLAG) { fl[kk] = jc | (code << 27); if (SINGLE) s_mimask[wave][kk] = (uint8_t)imask; }
                } else if (pass) {
                    uint32_t kk = np + __popcll(bp & lt_mask);
                    if (!SINGLE) a.entries[ebase + nm_pad_total + kk] = ent;
                    else if (kk < PLAINCAP) s_plain[wave][kk] = ent;
                }
            }
            nm += __popcll(bm);
            np += __popcll(bp);
            if (MODE != LB_FILL && pass) npairs += __popc(imask);
        }
        WAVE_LDS_SYNC();
        nc = 0;
    };
    for (uint32_t q0 = 0; q0 < ncombo; q0 += 64) {
        // (A) this lane's column, population and z-image; its window of tiles [tA, tB)
        uint32_t tA = 0, tB = 0, code = 13u;
        const uint32_t q = q0 + (uint32_t)lane;
        if (q < ncombo) {
            const int kz = kz0 + (int)(q % (uint32_t)nkz);
            const int pop = (int)((q / (uint32_t)nkz) % (uint32_t)g.npop);
            const int iy = iy0 + (int)((q / (uint32_t)(nkz * g.npop)) % (uint32_t)nyr);
            const int ix = ix0 + (int)(q / (uint32_t)(nkz * g.npop * nyr));
            const int kx = g.per[0] ? floor_div(ix, g.ncx) : 0, ky = g.per[1] ? floor_div(iy, g.ncy) : 0;
            const int wx = ix - kx * g.ncx, wy = iy - ky * g.ncy;
            const float sx = (float)kx * g.len[0], sy = (float)ky * g.len[1], sz = (float)kz * g.len[2];
            // column slab distance in x and y (column interval, slightly widened)
            const float cxlo = mdx_col_lo(g, 0, wx) + sx - 1e-3f, cxhi = mdx_col_lo(g, 0, wx + 1) + sx + 1e-3f;
            const float gx = (wx == 0 || wx == g.ncx - 1) ? 0.f : gap(cxlo, cxhi, lo[0], hi[0]);
            const float cylo = mdx_col_lo(g, 1, wy) + sy - 1e-3f, cyhi = mdx_col_lo(g, 1, wy + 1) + sy + 1e-3f;
            const float gy = (wy == 0 || wy == g.ncy - 1) ? 0.f : gap(cylo, cyhi, lo[1], hi[1]);
            if (gx * gx + gy * gy < r2) {
                const uint32_t c2 = (uint32_t)((pop * g.ncx + wx) * g.ncy + wy);
                const uint32_t t0 = a.tile_start[c2], t1 = a.tile_start[c2 + 1];
                code = (uint32_t)((kx + 1) + 3 * (ky + 1) + 9 * (kz + 1));
                tA = t0; tB = t1;
                // A column's tiles are consecutive z-ranges of its z-sorted atoms: only the tiles that hold
                // atoms of the z-bins within r of this tile can contribute (a 217 A column is 25 tiles tall,
                // the window ~4), so look the window up in the cell table instead of testing every cluster.
                if (r < 1.0e30f && t1 > t0) {
                    int zb0 = (int)floorf((lo[2] - r - sz - g.lo[2]) * g.inv_zbin) - 1;
                    int zb1 = (int)floorf((hi[2] + r - sz - g.lo[2]) * g.inv_zbin) + 1;
                    if (zb1 < 0 || zb0 >= g.nzb) tB = tA;
                    else {
                        zb0 = max(zb0, 0); zb1 = min(zb1, g.nzb - 1);
                        const uint32_t* cs = a.cell_start + (size_t)c2 * g.nzb;
                        const uint32_t aA = cs[zb0] - cs[0], aB = cs[zb1 + 1] - cs[0];
                        if (aB <= aA) tB = tA;
                        else { tA = t0 + aA / MDX_TILE; tB = min(t1, t0 + (aB + MDX_TILE - 1) / MDX_TILE); }
                    }
                }
            }
        }
        // compact the windows' tiles into the candidate buffer, one tile per lane and round.  Half list: a tile pair has one
        // owner, decided by the parity of I + J - tested here, it spares the non-owner the bounding-box pass altogether
        for (uint32_t k = 0; __any(tA + k < tB); ++k) {
            const uint32_t Jt = tA + k;
            const bool mine = Jt < tB && (!a.half || Jt == t || ((t < Jt) == (((t + Jt) & 1u) == 0u)));
            const unsigned long long bc = __ballot(mine);
            if (mine) cand[nc + __popcll(bc & lt_mask)] = Jt | (code << 27);
            nc += __popcll(bc);
            if (nc > LB_CAND - 64) drain();
        }
    }
    drain();
    const uint32_t nm_pad = (nm + 7) & ~7u, np_pad = (np + 7) & ~7u;
    if (nm > LB_MAXFLAG) atomicOr(a.err, SINGLE ? 16u : 2u);
    if (MODE != LB_FILL) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) npairs += __shfl_xor(npairs, m);
    }
    if (MODE == LB_COUNT) {
        if (lane == 0) {
            atomicAdd(pc_slot(a.pair_count, t, 0), (unsigned long long)npairs);
            a.counts[t].n_masked = nm_pad;
            a.counts[t].n_plain = np_pad;
            a.entry_cnt[t] = nm_pad + np_pad;
            a.mchunk_cnt[t] = nm_pad >> 3;
        }
        return;
    }
    uint32_t mbase = 0;
    if (SINGLE) {
        // claim this tile's slice of the entry array and of the mask array, then drain the LDS buffers into it
        bool fits = nm <= LB_MAXFLAG && np <= PLAINCAP;
        if (!fits) atomicOr(a.err, 16u);
        // A returning atomic on one address is served every ~35 ns (tools/ubench/grid_barrier.hip): two per tile on one
        // pair of cursors made 16 k tiles queue for 0.46 ms - the whole duration of this kernel at 1 M atoms, whatever its
        // occupancy or tile order.  Regions give the claims n_regions lines to land on.
        const uint32_t reg = t & (a.n_regions - 1u);
        unsigned long long claim = 0ull;
        if (lane == 0 && fits)
            claim = atomicAdd(a.cursors + (size_t)reg * 16, ((unsigned long long)(nm_pad >> 3) << 32) | (unsigned long long)(nm_pad + np_pad));
        const uint32_t eoff = __shfl((uint32_t)claim, 0), moff = __shfl((uint32_t)(claim >> 32), 0);
        ebase = reg * a.region_cap_e + eoff; mbase = reg * a.region_cap_m + moff;
        if (fits && (eoff + nm_pad + np_pad > a.region_cap_e || moff + (nm_pad >> 3) > a.region_cap_m)) {
            atomicOr(a.err, 32u);      // the host falls back to count + fill, which sizes the arrays afresh
            fits = false;
        }
        if (lane == 0) {
            a.counts[t].n_masked = fits ? nm_pad : 0u;
            a.counts[t].n_plain = fits ? np_pad : 0u;
            a.entry_off[t] = fits ? ebase : 0u;
            a.mchunk_off[t] = fits ? mbase : 0u;
            if (fits) atomicAdd(pc_slot(a.pair_count, t, 0), (unsigned long long)npairs);
        }
        if (!fits) return;
        nm_pad_total = nm_pad;
        WAVE_LDS_SYNC();
        for (uint32_t k = lane; k < nm; k += 64)
            a.entries[ebase + k] = make_uint2(fl[k] & 0x7FFFFFFu, (fl[k] >> 27) | ((uint32_t)s_mimask[wave][k] << 8));
        for (uint32_t k = lane; k < np; k += 64) a.entries[ebase + nm_pad_total + k] = s_plain[wave][k];
    }
    // pad both runs with null entries
    const uint2 null_ent = make_uint2(a.null_cluster, 13u);
    if (lane < (int)(nm_pad - nm)) a.entries[ebase + nm + lane] = null_ent;
    if (lane < (int)(np_pad - np)) a.entries[ebase + nm_pad_total + np + lane] = null_ent;
    WAVE_LDS_SYNC();

    // interaction masks of the masked run: bit (8*e + jj) of lane i <=> i interacts with atom jj
    // of the chunk's e-th entry.
    const uint32_t nmc = min(nm, (uint32_t)LB_MAXFLAG);
    if (!SINGLE) mbase = a.mchunk_off[t];
    for (uint32_t c = 0; c < (nm_pad >> 3); ++c) {
        unsigned long long m = 0ull;
        uint32_t jcs[8], codes13 = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint32_t idx = c * 8 + e;
            jcs[e] = (idx < nmc) ? (fl[idx] & 0x7FFFFFFu) : MDX_INVALID;
            if (a.half && idx < nmc && (fl[idx] >> 27) == 13u) codes13 |= 1u << e;
            if (idx < nmc) m |= 0xFFull << (8 * e);
        }
        if (myo == MDX_INVALID) {
            m = 0ull;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (jcs[e] == (myslot >> 3)) {
                    // self pair; in a half list the cluster's pair with itself keeps only j > i
                    const unsigned long long gone = ((codes13 >> e) & 1u) ? ((2ull << (myslot & 7)) - 1ull)
                                                                          : (1ull << (myslot & 7));
                    m &= ~(gone << (8 * e));
                }
#pragma unroll
            for (int q = 0; q < LB_SPC; ++q) {
                const uint32_t sp = spc[q];
                if (sp == MDX_INVALID) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (jcs[e] == (sp >> 3)) m &= ~(1ull << (8 * e + (sp & 7)));
            }
            for (uint32_t k = eb + LB_SPC; k < ee; ++k) {
                uint32_t sp = a.slot_of[a.excl_idx[k]];
                if (sp == MDX_INVALID) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (jcs[e] == (sp >> 3)) m &= ~(1ull << (8 * e + (sp & 7)));
            }
        }
        if (a.mask_layout == 2) {
            // cluster-masked kernel: lane (ii = lane&7, jj = lane>>3) needs, for entry e and
            // i-cluster ci, the bit "i-atom ci*8+ii interacts with j-atom jj of entry e"
            const int ii = lane & 7, jj = lane >> 3;
            unsigned long long mq = 0ull;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const unsigned long long mp = __shfl(m, ci * 8 + ii);
#pragma unroll
                for (int e = 0; e < 8; ++e) mq |= ((mp >> (8 * e + jj)) & 1ull) << (8 * e + ci);
            }
            m = mq;
        }
        a.masks[(size_t)(mbase + c) * 64 + lane] = m;
    }
}

// ================================================================================================
// Exact pruning of the cluster-pair masks.  The list kernel accepts (i-cluster, j-cluster) on bounding
// boxes; bricks of 8 atoms rarely fill their corners, and at rc 10 + skin 2 about 23 % of the accepted
// cluster pairs have NO atom pair inside the list radius - each of them a wasted 64-lane evaluation per
// step until the next rebuild.  This pass walks the finished list with the pair kernel's own lane mapping
// (lane = i-atom ii of every i-cluster x j-atom jj of the entry), so one cluster pair costs one distance
// test per lane and a ballot, and clears the imask bits of pairs without any atom pair inside r.  Entries of
// the plain run whose mask empties are squeezed out (the run is compacted in place and re-padded); entries of
// the masked run keep their position, which addresses their exclusion masks.  One wave per tile.
// ================================================================================================
#ifndef PRUNE_TRIES
#define PRUNE_TRIES 1         // quick-accept tries per cluster pair (measured at 1 M atoms: 0 -> 283 us, 1 -> 278, 3 -> 291, 6 -> 314)
#endif
// rb_ctl (fused rebuild; may be null): [0] the tile count, still on the device when this launch (for an upper bound of tiles)
// is enqueued, [1] the first ghost tile.  tile_int (may be null; half-shell decomposed handle: owned and ghost atoms sit in
// separate column populations, so every tile from rb_ctl[1] on is all-ghost): 1 <=> the tile is owned and no entry that
// SURVIVES this pass names a ghost cluster - everything its pair evaluation reads is owned by this rank, it may run before
// the halo message has arrived.  (It was a pass of its own over the finished list, tile_class_kernel, + a scan + a host copy.)
// INNER (round 6, the one-wave-per-tile class of a single-device handle): the same pass also derives the INNER list of the dual
// pair list - cluster pairs with an atom pair inside cutoff + inner_skin at the positions of this rebuild, which are the positions
// the force call behind the rebuild evaluates (the path accumulators start from zero there) - in the layout the pair kernel's own
// pruning pass writes for one wave per tile (masked run in place, plain run compacted from chunk n_masked / 8 on, loop bound in
// inner_nch[8 t]).  The force call behind a rebuild then WALKS the inner list instead of being a pruning pass over the Verlet
// list: 569 -> 418 us at 1 M atoms, for one more ballot per cluster pair here.  The quick accept is off in this flavour (every
// surviving cluster pair needs its distances for the second test anyway).
struct PruneInner { uint2* entries_in; uint32_t* inner_nch; float rin2; WptRule rule; };
template <bool INNER>
__global__ __launch_bounds__(256) void prune_list_kernel(uint32_t T, float r2, float shx, float shy, float shz,
                                                         const float4* __restrict__ posq, ListCounts* __restrict__ counts,
                                                         const uint32_t* __restrict__ entry_off, uint2* __restrict__ entries,
                                                         uint32_t null_cluster, unsigned long long* __restrict__ pair_count,
                                                         const uint32_t* __restrict__ rb_ctl, uint32_t* __restrict__ tile_int,
                                                         const uint8_t* __restrict__ cl_kind, PruneInner pin) {
    __shared__ float4 s_j[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t first_ghost_cluster = 0xFFFFFFFFu;
    if (rb_ctl) {
        T = __builtin_amdgcn_readfirstlane(rb_ctl[0]); null_cluster = T * MDX_CL_PER_TILE;
        first_ghost_cluster = __builtin_amdgcn_readfirstlane(rb_ctl[1]) * MDX_CL_PER_TILE;
    }
    // (the pair kernel reads the inner list in the layout of ITS waves-per-tile class: only the one-wave class is written here, and
    // the host applies the same rule to the same tile count when it decides what the force call behind the rebuild is)
    const bool inner_on = INNER && mdx_wpt_rule(pin.rule, T) == 1;
    bool ghost_hit = false;
    // one wave per tile (the plain run is compacted in place); a contiguous eighth of the tiles per XCD, as in the pair
    // kernel: the j-atoms a tile's list names are its spatial neighbours, and each XCD has its own L2
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t t = MDX_XCD_SWIZZLE ? ((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * 4 + wave : blockIdx.x * 4 + wave;
    if (t >= T) return;
    const int ii = lane & 7, jj = lane >> 3;
    float xi[8], yi[8], zi[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const float4 p = posq[(size_t)t * MDX_TILE + ci * MDX_CLUSTER + ii];
        xi[ci] = p.x; yi[ci] = p.y; zi[ci] = p.z;
    }
    // Quick accept before the 64-lane tests: lane (ci = lane & 7, e = lane >> 3) tries a fixed atom pair of cluster
    // pair (i-cluster ci, entry e); one within r settles that cluster pair for 7 instructions per CHUNK instead of 7 per
    // cluster pair.  What the try leaves open gets the exact test - the kept set is the same either way.  (The counters
    // say the pass is bound by VALU + SALU issue - 108 M + 107 M instructions per launch, the scalar half being the
    // ballot / branch bookkeeping per cluster pair - so more tries cost what they save.)
    constexpr int NTRY = INNER ? 0 : PRUNE_TRIES;
    constexpr int TI[6] = {0, 7, 3, 0, 7, 4}, TJ[6] = {0, 7, 4, 7, 0, 3};
    float4 rep[NTRY ? NTRY : 1];
#pragma unroll
    for (int k = 0; k < NTRY; ++k) rep[k] = posq[(size_t)t * MDX_TILE + (lane & 7) * MDX_CLUSTER + TI[k]];
    // interaction kinds (cl_kind, null: every cluster pair may interact): which i-clusters hold an atom with a Lennard-Jones
    // well / a charged atom.  A cluster pair with no kind in common - oxygens of a four-site water against hydrogens and M sites -
    // has no force between any of its atoms and leaves the list whatever the distances say.
    uint32_t ilj = 0xFFu, iq = 0xFFu;
    if (cl_kind) {
        const uint32_t k = lane < MDX_CL_PER_TILE ? cl_kind[(size_t)t * MDX_CL_PER_TILE + lane] : 0u;
        ilj = (uint32_t)__ballot((k & 1u) != 0u) & 0xFFu; iq = (uint32_t)__ballot((k & 2u) != 0u) & 0xFFu;
    }
    const ListCounts cnt = counts[t];
    const uint32_t e0 = entry_off[t], nmc = cnt.n_masked >> 3, nchunks = (cnt.n_masked + cnt.n_plain) >> 3;
    uint32_t kept = 0, wcur = 0;                   // wcur: plain entries written back so far
    uint32_t wcur_in = 0;                          // (INNER) ... and into the inner list
    // two-deep prefetch, as in the pair kernel: the entries of chunk c + 2 and the atoms of chunk c + 1 are in flight while
    // chunk c is tested (one-deep, the atom load waited for the entry load it depends on at the top of every chunk)
    uint2 ent_n = make_uint2(null_cluster, 13u), ent_nn = make_uint2(null_cluster, 13u);
    float4 pj_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nchunks) { ent_n = entries[e0 + (lane >> 3)]; pj_n = posq[(size_t)ent_n.x * MDX_CLUSTER + (lane & 7)]; }
    if (nchunks > 1) ent_nn = entries[e0 + 8 + (lane >> 3)];
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint2 ent = ent_n;
        float4 pj = pj_n;
        if (c + 1 < nchunks) {
            ent_n = ent_nn;
            pj_n = posq[(size_t)ent_n.x * MDX_CLUSTER + (lane & 7)];
            if (c + 2 < nchunks) ent_nn = entries[e0 + (c + 2) * 8 + (lane >> 3)];
        }
        const uint32_t code = ent.y & 31u;
        pj.x += (float)((int)(code % 3u) - 1) * shx;
        pj.y += (float)((int)((code / 3u) % 3u) - 1) * shy;
        pj.z += (float)((int)(code / 9u) - 1) * shz;
        s_j[wave][lane] = pj;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
        uint32_t newy = ent.y;                     // lanes 8e .. 8e+7 hold entry e
        if (cl_kind) {                             // i-clusters that share a kind with this lane's entry
            const uint32_t kj = ent.x < null_cluster ? cl_kind[ent.x] : 0u;
            newy = (ent.y & 0xFFu) | (ent.y & ((((kj & 1u) ? ilj : 0u) | ((kj & 2u) ? iq : 0u)) << 8));
        }
        const uint32_t y_in = newy;
        uint32_t newy_in = newy & 0xFFu;           // (INNER) this lane's entry word with the inner mask
        unsigned long long acc = 0ull;             // bit e * 8 + ci: cluster pair (ci, e) has an atom pair inside r for sure
#pragma unroll
        for (int k = 0; k < NTRY; ++k) {
            const float4 q = s_j[wave][(lane & ~7) + TJ[k]];
            const float dx = rep[k].x - q.x, dy = rep[k].y - q.y, dz = rep[k].z - q.z;
            acc |= __ballot(dx * dx + dy * dy + dz * dz < r2);
        }
#pragma unroll 2
        for (int e = 0; e < 8; ++e) {
            const float4 q = s_j[wave][e * 8 + jj];
            const uint32_t y = __builtin_amdgcn_readlane(y_in, e * 8);
            const uint32_t sure = (uint32_t)(acc >> (e * 8)) & 0xFFu;
            uint32_t any = sure, anyin = 0u;
            // only the i-clusters the bounding-box test let through (~5 of 8) and the quick accept left open
            // (all eight, branch-free: 357 us per rebuild at 1 M atoms; this: 278 us)
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (!(((y >> 8) & ~sure) & (1u << ci))) continue;
                const float dx = xi[ci] - q.x, dy = yi[ci] - q.y, dz = zi[ci] - q.z;
                const float d2 = dx * dx + dy * dy + dz * dz;
                const bool hit = __ballot(d2 < r2) != 0ull;
                any |= (hit ? 1u : 0u) << ci;
                if (INNER && hit) anyin |= (__ballot(d2 < pin.rin2) != 0ull ? 1u : 0u) << ci;
            }
            const uint32_t im = (y >> 8) & any;
            kept += __popc(im & 0xFFu);
            if ((lane >> 3) == e) newy = (y & 0xFFu) | ((im & 0xFFu) << 8);
            if (INNER && (lane >> 3) == e) newy_in = (y & 0xFFu) | ((((y >> 8) & anyin) & 0xFFu) << 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
        ghost_hit |= ((newy >> 8) & 0xFFu) != 0u && ent.x >= first_ghost_cluster && ent.x < null_cluster;
        if (c < nmc) {
            // masked run: per-lane exclusion masks are addressed by chunk position, entries stay where they are
            if ((lane & 7) == 0 && newy != ent.y) entries[e0 + c * 8 + (lane >> 3)].y = newy;
            if (inner_on && (lane & 7) == 0) pin.entries_in[e0 + c * 8 + (lane >> 3)] = make_uint2(ent.x, newy_in);
        } else {
            // plain run: survivors move down (write cursor <= read position, and the next chunk is already in registers)
            const bool alive = (lane & 7) == 0 && ((newy >> 8) & 0xFFu) != 0u;
            const unsigned long long bal = __ballot(alive);
            if (alive) {
                const uint32_t rank = __popcll(bal & ((1ull << lane) - 1ull));
                entries[e0 + cnt.n_masked + wcur + rank] = make_uint2(ent.x, newy);
            }
            wcur += __popcll(bal);
            if (inner_on) {
                const bool alive_in = (lane & 7) == 0 && ((newy_in >> 8) & 0xFFu) != 0u;
                const unsigned long long bin = __ballot(alive_in);
                if (alive_in) pin.entries_in[e0 + cnt.n_masked + wcur_in + __popcll(bin & ((1ull << lane) - 1ull))] = make_uint2(ent.x, newy_in);
                wcur_in += __popcll(bin);
            }
        }
    }
    const uint32_t np_pad = (wcur + 7u) & ~7u;
    if (lane < (int)(np_pad - wcur)) entries[e0 + cnt.n_masked + wcur + lane] = make_uint2(null_cluster, 13u);
    if (inner_on) {
        const uint32_t nin_pad = (wcur_in + 7u) & ~7u;
        if (lane < (int)(nin_pad - wcur_in)) pin.entries_in[e0 + cnt.n_masked + wcur_in + lane] = make_uint2(null_cluster, 13u);
        if (lane == 0) pin.inner_nch[(size_t)t * 8] = nmc + (nin_pad >> 3);
    }
    const bool any_ghost = __any(ghost_hit);
    if (lane == 0) {
        counts[t].n_plain = np_pad;
        if (kept) atomicAdd(pc_slot(pair_count, t, 1), (unsigned long long)kept);
        if (tile_int) tile_int[t] = (!any_ghost && t * MDX_CL_PER_TILE < first_ghost_cluster) ? 1u : 0u;
    }
}

// The same pass for systems of a few hundred tiles, W waves per tile: one wave per tile leaves such a launch at the
// latency of its longest list (82 us at 23 k atoms, 370 tiles on 1024 SIMDs).  Wave w takes chunks w, w + W, ... and
// writes the trimmed masks in place; after a workgroup barrier wave 0 squeezes the emptied entries out of the plain
// run, 64 entries per step.  Same kept set, same layout as prune_list_kernel.
template <int W>
__global__ __launch_bounds__(W * 64) void prune_list_mw_kernel(uint32_t T, float r2, float shx, float shy, float shz,
                                                               const float4* __restrict__ posq, ListCounts* __restrict__ counts,
                                                               const uint32_t* __restrict__ entry_off, uint2* __restrict__ entries,
                                                               uint32_t null_cluster, unsigned long long* __restrict__ pair_count,
                                                               const uint32_t* __restrict__ rb_ctl, uint32_t* __restrict__ tile_int,
                                                               const uint8_t* __restrict__ cl_kind) {
    __shared__ float4 s_j[W][64];
    __shared__ uint32_t s_ghost;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t first_ghost_cluster = 0xFFFFFFFFu;
    if (rb_ctl) {      // (see prune_list_kernel)
        T = __builtin_amdgcn_readfirstlane(rb_ctl[0]); null_cluster = T * MDX_CL_PER_TILE;
        first_ghost_cluster = __builtin_amdgcn_readfirstlane(rb_ctl[1]) * MDX_CL_PER_TILE;
    }
    const uint32_t t = blockIdx.x;
    if (t >= T) return;
    if (threadIdx.x == 0) s_ghost = 0u;
    if (tile_int) __syncthreads();
    bool ghost_hit = false;
    const int ii = lane & 7, jj = lane >> 3;
    float xi[8], yi[8], zi[8];
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        const float4 p = posq[(size_t)t * MDX_TILE + ci * MDX_CLUSTER + ii];
        xi[ci] = p.x; yi[ci] = p.y; zi[ci] = p.z;
    }
    const float4 rep = posq[(size_t)t * MDX_TILE + (lane & 7) * MDX_CLUSTER];      // quick accept: atom 0 of i-cluster (lane & 7)
    // interaction kinds (cl_kind, null: every cluster pair may interact): which i-clusters hold an atom with a Lennard-Jones
    // well / a charged atom.  A cluster pair with no kind in common - oxygens of a four-site water against hydrogens and M sites -
    // has no force between any of its atoms and leaves the list whatever the distances say.
    uint32_t ilj = 0xFFu, iq = 0xFFu;
    if (cl_kind) {
        const uint32_t k = lane < MDX_CL_PER_TILE ? cl_kind[(size_t)t * MDX_CL_PER_TILE + lane] : 0u;
        ilj = (uint32_t)__ballot((k & 1u) != 0u) & 0xFFu; iq = (uint32_t)__ballot((k & 2u) != 0u) & 0xFFu;
    }
    const ListCounts cnt = counts[t];
    const uint32_t e0 = entry_off[t], nchunks = (cnt.n_masked + cnt.n_plain) >> 3;
    uint32_t kept = 0;
    uint2 ent_n = make_uint2(null_cluster, 13u);
    float4 pj_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((uint32_t)wave < nchunks) { ent_n = entries[e0 + wave * 8 + (lane >> 3)]; pj_n = posq[(size_t)ent_n.x * MDX_CLUSTER + (lane & 7)]; }
    for (uint32_t c = wave; c < nchunks; c += W) {
        const uint2 ent = ent_n;
        float4 pj = pj_n;
        if (c + W < nchunks) {
            ent_n = entries[e0 + (c + W) * 8 + (lane >> 3)];
            pj_n = posq[(size_t)ent_n.x * MDX_CLUSTER + (lane & 7)];
        }
        const uint32_t code = ent.y & 31u;
        pj.x += (float)((int)(code % 3u) - 1) * shx;
        pj.y += (float)((int)((code / 3u) % 3u) - 1) * shy;
        pj.z += (float)((int)(code / 9u) - 1) * shz;
        s_j[wave][lane] = pj;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
        uint32_t newy = ent.y;
        if (cl_kind) {                             // (see prune_list_kernel)
            const uint32_t kj = ent.x < null_cluster ? cl_kind[ent.x] : 0u;
            newy = (ent.y & 0xFFu) | (ent.y & ((((kj & 1u) ? ilj : 0u) | ((kj & 2u) ? iq : 0u)) << 8));
        }
        const uint32_t y_in = newy;
        unsigned long long acc;
        {
            const float4 q = s_j[wave][lane & ~7];
            const float dx = rep.x - q.x, dy = rep.y - q.y, dz = rep.z - q.z;
            acc = __ballot(dx * dx + dy * dy + dz * dz < r2);
        }
#pragma unroll 2
        for (int e = 0; e < 8; ++e) {
            const float4 q = s_j[wave][e * 8 + jj];
            const uint32_t y = __builtin_amdgcn_readlane(y_in, e * 8);
            const uint32_t sure = (uint32_t)(acc >> (e * 8)) & 0xFFu;
            uint32_t any = sure;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (!(((y >> 8) & ~sure) & (1u << ci))) continue;
                const float dx = xi[ci] - q.x, dy = yi[ci] - q.y, dz = zi[ci] - q.z;
                any |= (__ballot(dx * dx + dy * dy + dz * dz < r2) != 0ull ? 1u : 0u) << ci;
            }
            const uint32_t im = (y >> 8) & any;
            kept += __popc(im & 0xFFu);
            if ((lane >> 3) == e) newy = (y & 0xFFu) | ((im & 0xFFu) << 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
        if ((lane & 7) == 0 && newy != ent.y) entries[e0 + c * 8 + (lane >> 3)].y = newy;
        ghost_hit |= ((newy >> 8) & 0xFFu) != 0u && ent.x >= first_ghost_cluster && ent.x < null_cluster;
    }
    if (lane == 0 && kept) atomicAdd(pc_slot(pair_count, t, 1), (unsigned long long)kept);
    if (tile_int && __any(ghost_hit) && lane == 0) atomicOr(&s_ghost, 1u);
    __threadfence_block();
    __syncthreads();
    if (wave != 0) return;
    if (tile_int && lane == 0) tile_int[t] = (s_ghost == 0u && t * MDX_CL_PER_TILE < first_ghost_cluster) ? 1u : 0u;
    // plain run: survivors move down (reads of a 64-entry step are complete - the ballot needs them - before its writes,
    // and the write cursor never passes the read position)
    uint32_t wcur = 0;
    uint2* const plain = entries + e0 + cnt.n_masked;
    for (uint32_t base = 0; base < cnt.n_plain; base += 64) {
        const uint32_t idx = base + (uint32_t)lane;
        const uint2 ent = idx < cnt.n_plain ? plain[idx] : make_uint2(null_cluster, 13u);
        const bool alive = ((ent.y >> 8) & 0xFFu) != 0u;
        const unsigned long long bal = __ballot(alive);
        if (alive) plain[wcur + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = ent;
        wcur += (uint32_t)__popcll(bal);
    }
    const uint32_t np_pad = (wcur + 7u) & ~7u;
    if (lane < (int)(np_pad - wcur)) plain[wcur + lane] = make_uint2(null_cluster, 13u);
    if (lane == 0) counts[t].n_plain = np_pad;
}

// ================================================================================================
// neighbour-list extraction (parity / debugging API): canonical fp32 distances
// ================================================================================================
__device__ __forceinline__ float r2_canonical(float4 pi, float4 pj, const float* L, const int* per) {
    float dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
    if (per[0]) dx = dx - rintf(dx / L[0]) * L[0];
    if (per[1]) dy = dy - rintf(dy / L[1]) * L[1];
    if (per[2]) dz = dz - rintf(dz / L[2]) * L[2];
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

template <bool FILL>
__global__ __launch_bounds__(256) void extract_neighbors_kernel(
    uint32_t T, GridParams g, float rl2, int half, const uint32_t* __restrict__ entry_off, const ListCounts* __restrict__ counts,
    const uint2* __restrict__ entries, const uint32_t* __restrict__ orig_of, const float4* __restrict__ ref,
    uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, uint32_t* __restrict__ cursor,
    uint32_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const uint32_t slot = t * MDX_TILE + lane;
    const uint32_t oi = orig_of[slot];
    const float4 pi = ref[slot];
    uint32_t k = 0;
    uint32_t* row = (FILL && !half && oi != MDX_INVALID) ? idx + off[oi] : nullptr;
    const uint32_t e_end = entry_off[t] + counts[t].n_masked + counts[t].n_plain;   // (tiles are not stored in tile order)
    for (uint32_t e = entry_off[t]; e < e_end; ++e) {
        const uint32_t jc = entries[e].x;
        for (int jj = 0; jj < MDX_CLUSTER; ++jj) {
            const uint32_t js = jc * MDX_CLUSTER + jj;
            const uint32_t oj = orig_of[js];
            if (oj == MDX_INVALID || oi == MDX_INVALID || oj == oi) continue;
            if (r2_canonical(pi, ref[js], g.len, g.per) < rl2) {
                if (half) {
                    // a half list holds the pair on one side only: emit both directions (the host
                    // sorts the rows and drops duplicates)
                    if (FILL) {
                        idx[off[oi] + atomicAdd(&cursor[oi], 1u)] = oj;
                        idx[off[oj] + atomicAdd(&cursor[oj], 1u)] = oi;
                    } else {
                        atomicAdd(&cnt[oi], 1u); atomicAdd(&cnt[oj], 1u);
                    }
                } else {
                    if (FILL) row[k] = oj;
                    ++k;
                }
            }
        }
    }
    if (!FILL && !half && oi != MDX_INVALID) cnt[oi] = k;
}

// ================================================================================================
// host orchestration
// ================================================================================================
static int alloc_dev(void** p, size_t bytes) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    return MDX_OK;
}
#define ALLOC(ptr, count) MDX_TRY(alloc_dev((void**)&(ptr), sizeof(*(ptr)) * (size_t)(count)))

static int setup_grid(mdx_handle* h) {
    GridParams& g = h->grid;
    const uint32_t N = h->n_local;
    bool need_bbox = false;
    for (int d = 0; d < 3; ++d) {
        g.per[d] = h->per[d];
        if (h->per[d]) { g.lo[d] = h->box_lo[d]; g.len[d] = h->box_hi[d] - h->box_lo[d]; }
        else if (h->have_local_bounds) { g.lo[d] = h->local_lo[d]; g.len[d] = std::max(h->local_hi[d] - h->local_lo[d], 1.0f); }
        else need_bbox = true;
    }
    if (need_bbox) {
        // bounding box of the current caller-order positions (vacuum systems are small)
        std::vector<float4> hp(N);
        HIP_TRY(hipMemcpyAsync(hp.data(), h->d.pos_orig, sizeof(float4) * N, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t i = 0; i < N; ++i) {
            const float v[3] = {hp[i].x, hp[i].y, hp[i].z};
            for (int d = 0; d < 3; ++d) {
                if (!std::isfinite(v[d])) { mdx_set_error("non-finite position"); return MDX_ENAN; }
                lo[d] = std::min(lo[d], v[d]); hi[d] = std::max(hi[d], v[d]);
            }
        }
        for (int d = 0; d < 3; ++d)
            if (!h->per[d]) { g.lo[d] = lo[d] - 0.01f; g.len[d] = std::max(hi[d] - lo[d] + 0.02f, 1.0f); }
    }
    const double vol = (double)g.len[0] * g.len[1] * g.len[2];
    const double rho = std::max((double)N / vol, 1e-4);
    double s = std::cbrt(64.0 / rho);            // cubic 64-atom bricks at the mean density
    s = std::min(std::max(s, 3.0), 24.0);
    g.ncx = std::max(1, (int)std::floor(g.len[0] / s));
    g.ncy = std::max(1, (int)std::floor(g.len[1] / s));
    while ((double)g.ncx * g.ncy > 4.0e6) { g.ncx = std::max(1, g.ncx / 2); g.ncy = std::max(1, g.ncy / 2); }
    g.inv_col[0] = (float)(g.ncx / (double)g.len[0]);
    g.inv_col[1] = (float)(g.ncy / (double)g.len[1]);
    {   // piecewise columns in the cut dimensions of a half-shell decomposed handle (GridParams; MDX_GRID_PIECEWISE=0: uniform, A/B)
        static const bool pw_env = [] { const char* e = std::getenv("MDX_GRID_PIECEWISE"); return !(e && e[0] == '0'); }();
        for (int d = 0; d < 2; ++d) {
            g.pw[d] = 0; g.nh[d] = 0; g.nbk[d] = 0; g.b_lo[d] = g.b_hi[d] = 0.f; g.inv_wh[d] = g.inv_wb[d] = 0.f;
            float blo = 0.f, bhi = 0.f;
            if (!pw_env || h->per[d] || !h->have_local_bounds || !mdx_dd_half_shell(h) || !mdx_dd_brick_bounds(h, d, &blo, &bhi)) continue;
            const double wh = (double)blo - (double)g.lo[d], wb = (double)bhi - (double)blo, wh2 = (double)g.lo[d] + (double)g.len[d] - (double)bhi;
            if (!(wh > 0.5 && wb > 0.5 && std::fabs(wh - wh2) < 1e-2 * wh)) continue;
            g.pw[d] = 1; g.b_lo[d] = blo; g.b_hi[d] = bhi;
            g.nh[d] = std::max(1, (int)std::lround(wh / s)); g.nbk[d] = std::max(1, (int)std::lround(wb / s));
            g.inv_wh[d] = (float)(g.nh[d] / wh); g.inv_wb[d] = (float)(g.nbk[d] / wb);
            (d == 0 ? g.ncx : g.ncy) = 2 * g.nh[d] + g.nbk[d];
            g.inv_col[d] = (float)((d == 0 ? g.ncx : g.ncy) / (double)g.len[d]);      // (the mean width: sizes the z-bins below)
        }
    }
    const double col_area = ((double)g.len[0] / g.ncx) * ((double)g.len[1] / g.ncy);
    double zh = 6.0 / (rho * col_area);          // ~6 atoms per fine cell
    zh = std::min(std::max(zh, 0.25), (double)g.len[2]);
    long nzb = std::max(1L, (long)std::ceil(g.len[2] / zh));
    g.npop = mdx_dd_half_shell(h) ? 2 : 1;
    const long ncol = (long)g.ncx * g.ncy * g.npop;
    while (ncol * nzb > 48L * 1000 * 1000) nzb = (nzb + 1) / 2;
    g.nzb = (int)nzb;
    g.inv_zbin = (float)(g.nzb / (double)g.len[2]);
    const uint32_t new_ncol = (uint32_t)ncol, new_ncells = (uint32_t)(ncol * nzb);
    DeviceState& d = h->d;
    if (new_ncells > h->ncells || !d.cell_count) {
        ALLOC(d.cell_count, (size_t)new_ncells + 1);
        h->cell_count_clean = false;
        ALLOC(d.cell_start, (size_t)new_ncells + 1);
        ALLOC(d.cell_cursor, (size_t)new_ncells + 1);
    }
    if (new_ncol > h->ncol || !d.col_tiles) {
        ALLOC(d.col_tiles, (size_t)new_ncol + 1);
        ALLOC(d.tile_start, (size_t)new_ncol + 1);
    }
    h->ncol = new_ncol; h->ncells = new_ncells;
    {   // scratch of the exclusive scans: one partial per 2048 elements of the longest scanned array
        const size_t longest = std::max<size_t>((size_t)new_ncells + 1, (size_t)N + 64 * (size_t)new_ncol + 256);
        const size_t need = longest / SCAN_TILE + 64;
        if (need > h->cap_scan) { h->cap_scan = need; ALLOC(d.scan_tmp, need); }
    }
    // capacity in tiles: sum ceil(cnt/64) <= N/64 + ncol, + the null tile
    const uint32_t need_tiles = N / MDX_TILE + new_ncol + 2;
    if (need_tiles > h->cap_tiles) {
        // the slot-space arrays are about to be re-allocated (ALLOC does not copy): a handle whose dynamic state lives in them - the
        // fused rebuild skips the unsort - saves it into the caller-order staging first (mdx_rebuild then leaves the fused chain)
        if (h->in_slot_space && h->cap_tiles) MDX_TRY(mdx_unsort_state(h));
        h->cap_tiles = need_tiles;
        const size_t S = (size_t)need_tiles * MDX_TILE, NC = (size_t)need_tiles * MDX_CL_PER_TILE;
        ALLOC(d.posq, S); ALLOC(d.lj, S); ALLOC(d.vel, S); ALLOC(d.force, S); ALLOC(d.ref, S); ALLOC(d.posq_alt, S); ALLOC(d.path, S); ALLOC(d.dprune, S);
        ALLOC(d.force_b, S); ALLOC(d.force_c, S);
        ALLOC(d.orig_of, S); ALLOC(d.slot_flags, S); ALLOC(d.pme_force, S);
        ALLOC(d.role_cnt_s, S + 1); ALLOC(d.role_off_s, S + 1);
        ALLOC(d.tile_col, need_tiles);
        ALLOC(d.cl_lo, NC); ALLOC(d.cl_hi, NC); ALLOC(d.cl_kind, NC);
        ALLOC(d.list_counts, need_tiles);
        ALLOC(d.inner_nch, (size_t)need_tiles * 8);
        ALLOC(d.entry_cnt, (size_t)need_tiles + 1); ALLOC(d.entry_off, (size_t)need_tiles + 1);
        ALLOC(d.mchunk_cnt, (size_t)need_tiles + 1); ALLOC(d.mchunk_off, (size_t)need_tiles + 1);
    }
    return MDX_OK;
}

int mdx_unsort_state(mdx_handle* h) {
    if (!h->in_slot_space) return MDX_OK;
    hipLaunchKernelGGL(unsort_kernel, dim3(div_up(h->n_local, 256)), dim3(256), 0, h->stream, h->n_local, h->d.gid,
                       h->d.slot_of, h->d.posq, h->d.vel, h->d.pos_orig, h->d.vel_orig);
    HIP_TRY(hipGetLastError());
    h->in_slot_space = false;
    return MDX_OK;
}

int mdx_gather_to_orig(mdx_handle* h, const float4* slot_arr, float4* orig_arr) {
    hipLaunchKernelGGL(gather_orig_kernel, dim3(div_up(h->n_local, 256)), dim3(256), 0, h->stream, h->n_local,
                       h->d.gid, h->d.slot_of, slot_arr, orig_arr);
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

// The rebuild's host decisions hang on a few device words (tile count, error bits, list cursors, pair counts).  One
// thread copies them into pinned host memory the device can write: one 3 us launch per synchronisation point instead
// of two to four pageable hipMemcpyAsync calls of ~25 us each (the rebuild at 1 M atoms spent 0.3 of its 1.36 ms in
// those copies and the host wake-ups around them).
struct RbSrc { const uint32_t* p[4]; uint32_t n[4]; };
__global__ void readback_kernel(RbSrc s, volatile uint32_t* out, uint32_t seq) {
    uint32_t k = 0;
    for (int a = 0; a < 4; ++a)
        for (uint32_t i = 0; i < s.n[a]; ++i) out[k++] = s.p[a] ? s.p[a][i] : 0u;
    __threadfence_system();
    out[31] = seq;      // the host spins on this word (a stream synchronisation costs ~5 us more: tools/ubench/sync_latency.hip)
}
static int readback(mdx_handle* h, const RbSrc& s) {
    if (!h->h_rb) { HIP_TRY(hipHostMalloc((void**)&h->h_rb, sizeof(uint32_t) * 32, hipHostMallocDefault)); h->h_rb[31] = 0u; }
    static const bool spin_ok = [] { const char* e = std::getenv("MDX_CHUNK_SPIN"); return !(e && e[0] == '0'); }();
    const uint32_t seq = ++h->rb_seq;
    hipLaunchKernelGGL(readback_kernel, dim3(1), dim3(1), 0, h->stream, s, (volatile uint32_t*)h->h_rb, seq);
    if (!spin_ok || h->profile) { HIP_TRY(hipStreamSynchronize(h->stream)); return MDX_OK; }
    volatile uint32_t* seq_word = (volatile uint32_t*)h->h_rb + 31;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *seq_word != seq; ++spins) {
        if ((spins & 0xFFFu) == 0xFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            HIP_TRY(hipStreamSynchronize(h->stream));       // (an error on the stream, or a very long queue: wait the ordinary way)
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return MDX_OK;
}

// totals of the single-pass build's region cursors (statistics): entries, masked chunks
__global__ void cursor_sum_kernel(const unsigned long long* __restrict__ cursors, uint32_t* __restrict__ out) {
    const unsigned long long c = cursors[(size_t)threadIdx.x * 16];
    uint32_t e = (uint32_t)c, m = (uint32_t)(c >> 32);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { e += __shfl_xor(e, k); m += __shfl_xor(m, k); }
    if (threadIdx.x == 0) { out[0] = e; out[1] = m; }
}

__global__ void pair_sum_kernel(unsigned long long* __restrict__ pc) {      // one wave
    unsigned long long a = pc[(size_t)threadIdx.x * PC_STRIDE], b = pc[(size_t)threadIdx.x * PC_STRIDE + 1];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { a += __shfl_xor(a, k); b += __shfl_xor(b, k); }
    if (threadIdx.x == 0) { pc[PC_TOTAL] = a; pc[PC_TOTAL + 1] = b; }
}

static float c_inner_skin(const mdx_config& c) { return c.inner_skin == 0.f ? 0.5f : c.inner_skin; }

// ================================================================================================
// Fused list rebuild (round 4).  The chain above is ~28 launches with three host synchronisation points (the tile count,
// the list totals, the interior-tile count of a decomposed handle); at 1 M atoms its small kernels are a third of the
// rebuild, and for one rank of an 8-GPU run - 200 k atoms - launch latency and host round trips are most of it (0.49 ms of
// which the two list kernels are 0.1).  Here the tile count never leaves the device until the end:
//   prep      (atom)   slot space -> caller order (positions, velocities), wrap, cell id, histogram - the atomicAdd's return
//                      value is the atom's arrival rank in its cell, so the scatter below needs no cursor
//   gridscan  (1 workgroup) exclusive scan of the cell histogram (which it leaves zeroed for the next rebuild), tiles per
//                      column, their scan, tile -> column, T; clears the list cursors, the statistics and the error bits
//   scatter, rank (atom)   cell members in arrival order, then in (z, atom id) order: deterministic
//   assign    (wave = tile, launched for an upper bound of tiles) the 64 atoms of a tile, bitonic sub-sort, slot-space arrays,
//                      cluster bounding boxes and the slots' bonded-role counts - assign + gather + bbox + role count in one
//   role scan (1 or 3 launches, length read on the device)
//   list      build_list_kernel<LB_SINGLE> as before, role fill in its extra workgroups; T read on the device
//   prune     as before; on a half-shell decomposed handle it also says which tiles are interior
//   finish    (1 workgroup) tiles by list length (inside the pair kernel's XCD ranges, or interior first), totals, and the
//                      one read-back into pinned host memory the host spins on
// = 9-11 launches, one host wait.  Anything unusual (first build, vacuum systems, overflow of the list arrays, more than a
// million cells, MDX_REBUILD_FUSED=0) takes the chain above.
// ================================================================================================
constexpr uint32_t LPT_BUCKETS = 129;      // tiles by list length: a counting sort over the chunk count (lpt_order_kernel below)
enum { RB_T = 0, RB_T_OWN = 1, RB_NONFINITE = 2, RB_N_INT = 3, RB_SCAN_N = 4, RB_WORDS = 8 };

template <bool UNSORT>
__global__ __launch_bounds__(256) void rb_prep_kernel(uint32_t N, GridParams g, const uint32_t* __restrict__ gid, const uint32_t* __restrict__ slot_of,
                                                      const float4* __restrict__ posq, const float4* __restrict__ vel,
                                                      float4* __restrict__ pos_orig, float4* __restrict__ vel_orig,
                                                      const uint8_t* __restrict__ lflag, uint32_t* __restrict__ cell_of,
                                                      uint32_t* __restrict__ cell_k, uint32_t* __restrict__ cell_count,
                                                      uint32_t* __restrict__ rb_ctl) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    float4 p;
    if (UNSORT) { const uint32_t sl = slot_of[gid[o]]; p = posq[sl]; vel_orig[o] = vel[sl]; }
    else p = pos_orig[o];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {
        rb_ctl[RB_NONFINITE] = 1u;
        p.x = g.lo[0]; p.y = g.lo[1]; p.z = g.lo[2];
    }
    if (g.per[0]) p.x = wrap1(p.x, g.lo[0], g.len[0]);
    if (g.per[1]) p.y = wrap1(p.y, g.lo[1], g.len[1]);
    if (g.per[2]) p.z = wrap1(p.z, g.lo[2], g.len[2]);
    pos_orig[o] = p;
    const int cx = min(g.ncx - 1, max(0, mdx_col_of(g, 0, p.x)));
    const int cy = min(g.ncy - 1, max(0, mdx_col_of(g, 1, p.y)));
    const int zb = min(g.nzb - 1, max(0, (int)((p.z - g.lo[2]) * g.inv_zbin)));
    const int pop = (g.npop > 1 && (lflag[o] & 1u)) ? 1 : 0;      // ghosts get column sets of their own
    const uint32_t cell = (uint32_t)(((pop * g.ncx + cx) * g.ncy + cy) * g.nzb + zb);
    cell_of[o] = cell;
    cell_k[o] = atomicAdd(&cell_count[cell], 1u);
}

struct GridScanArgs {
    uint32_t ncells, ncol, first_ghost_col; int nzb;
    uint32_t* cell_count; uint32_t* cell_start; uint32_t* tile_start; uint32_t* tile_col;
    uint32_t* rb_ctl; uint32_t* flags; unsigned long long* pair_count; uint32_t* list_cursors;
    ScanChain chain;
};
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN1_THREADS / 64; ++w) { const uint32_t x = s_wave[w]; if (w < wave) base += x; tot += x; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
__global__ __launch_bounds__(SCAN1_THREADS) void rb_gridscan_kernel(GridScanArgs a) {
    __shared__ uint32_t s_wave[SCAN1_THREADS / 64];
    __shared__ uint32_t s_bcast;
    // (the histogram has ncells + 1 words, the last one always zero: cell_start[ncells] is the atom count)
    const uint32_t n = a.ncells + 1u, w = blockIdx.x, w0 = w * SCAN1_WINDOW, nwin = gridDim.x;
    (void)scan_window_1024(a.cell_count + w0, a.cell_start + w0, min(SCAN1_WINDOW, n - w0),
                           [&](uint32_t tot) { return chained_carry(a.chain, w, tot, &s_bcast); }, s_wave, a.cell_count + w0);
    // every window's offsets must have landed before the columns are counted: the workgroups tick a generation-stamped word
    // each (release), the last one waits for all of them (acquire) and does the rest alone
    // (the hand-off forms of MI355X_MICROARCH.md, "inter-workgroup visibility": plain stores -> barrier -> one lane's agent
    // release -> drained -> relaxed agent flag; the consumer polls relaxed, then ONE agent acquire, barrier, plain loads)
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.chain.carry + 32 + w, (unsigned long long)a.chain.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (w + 1 != nwin) return;
    if (threadIdx.x < nwin) {
        while (__hip_atomic_load(a.chain.carry + 32 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)a.chain.gen) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // tiles per column and their running sum
    uint32_t run = 0;
    for (uint32_t c0 = 0; c0 < a.ncol; c0 += SCAN1_THREADS) {
        const uint32_t c = c0 + threadIdx.x;
        uint32_t nt = 0, tot;
        if (c < a.ncol)
            nt = (a.cell_start[(size_t)(c + 1) * a.nzb] - a.cell_start[(size_t)c * a.nzb] + MDX_TILE - 1) / MDX_TILE;
        const uint32_t ex = block_exclusive_scan_1024(nt, s_wave, &tot);
        if (c < a.ncol) {
            a.tile_start[c] = run + ex;
            for (uint32_t t = 0; t < nt; ++t) a.tile_col[run + ex + t] = c;
            if (c == a.first_ghost_col) a.rb_ctl[RB_T_OWN] = run + ex;
        }
        run += tot;
    }
    if (threadIdx.x == 0) {
        a.tile_start[a.ncol] = run;
        a.rb_ctl[RB_T] = run;
        if (a.first_ghost_col >= a.ncol) a.rb_ctl[RB_T_OWN] = run;
        a.rb_ctl[RB_SCAN_N] = (run + 1u) * MDX_TILE + 1u;
    }
    if (threadIdx.x < 4) a.flags[threadIdx.x] = 0u;
    for (uint32_t i = threadIdx.x; i < PC_TOTAL + 2u; i += SCAN1_THREADS) a.pair_count[i] = 0ull;
    for (uint32_t i = threadIdx.x; i < (uint32_t)LB_REGIONS * 32u + 2u; i += SCAN1_THREADS) a.list_cursors[i] = 0u;
}

__global__ __launch_bounds__(256) void rb_scatter_kernel(uint32_t N, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ cell_k,
                                                         const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ sorted_tmp) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    sorted_tmp[cell_start[cell_of[o]] + cell_k[o]] = o;
}

struct AssignArgs {
    const uint32_t* rb_ctl; int nzb;
    const uint32_t* tile_col; const uint32_t* tile_start; const uint32_t* cell_start; const uint32_t* sorted_orig;
    const float4* pos_orig; const float4* vel_orig; const uint32_t* gid; const uint8_t* lflag;
    const float* o_qs; const float2* o_lj; const float* o_invm;
    uint32_t* orig_of; uint32_t* slot_of;
    float4* posq; float2* lj; float4* vel; float4* ref; float4* force; uint8_t* slot_flags;
    float* path; float* dprune;      // path split (DeviceState): both start at zero with the new reference positions
    float4* cl_lo; float4* cl_hi;
    const uint32_t* role_off_o; uint32_t* role_cnt_s;     // null without bonded roles
    uint8_t* cl_kind; int kind_split;                     // interaction kinds per cluster; kind_split: clusters are formed per kind
};
__global__ __launch_bounds__(256) void rb_assign_kernel(AssignArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t T = __builtin_amdgcn_readfirstlane(a.rb_ctl[RB_T]);
    const uint32_t t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t > T) return;
    uint32_t o = MDX_INVALID;
    float x = FLT_MAX, y = FLT_MAX, z = FLT_MAX;
    if (t < T) {       // (tile T is the null tile: all dummies)
        const uint32_t c = a.tile_col[t];
        const uint32_t a0 = a.cell_start[(size_t)c * a.nzb] + (t - a.tile_start[c]) * MDX_TILE;
        const uint32_t aend = a.cell_start[(size_t)(c + 1) * a.nzb];
        const uint32_t p = a0 + lane;
        if (p < aend) { o = a.sorted_orig[p]; const float4 q = a.pos_orig[o]; x = q.x; y = q.y; z = q.z; }
        if (a.kind_split) {
            int group = 2;
            if (o != MDX_INVALID) {
                const uint32_t g = a.gid[o];
                const bool silent = (a.lflag[o] & 2u) != 0;
                group = (!silent && a.o_lj[g].y != 0.f && a.o_qs[g] == 0.f) ? 0 : 1;
            }
            kind_tile_order(x, y, z, o, group, lane);
        } else {
            // halves by y, quarters by x, eighths (clusters) by z
            bitonic_group<64>(y, x, z, o, lane);
            bitonic_group<32>(x, y, z, o, lane);
            bitonic_group<16>(z, x, y, o, lane);
        }
    }
    const uint32_t s = t * MDX_TILE + lane;
    a.orig_of[s] = o;
    float4 p, v; float2 l;
    uint8_t fl = 0;
    uint32_t nrole = 0;
    const bool valid = o != MDX_INVALID;
    if (valid) {
        const uint32_t g = a.gid[o];
        a.slot_of[g] = s;
        const uint8_t lf = a.lflag[o];
        const bool ghost = (lf & 1u) != 0, silent = (lf & 2u) != 0;   // (bit 1: kept only as the bonded partner of an owned atom)
        const float4 q = a.pos_orig[o], w = a.vel_orig[o];
        p = make_float4(q.x, q.y, q.z, silent ? 0.f : a.o_qs[g]);
        v = ghost ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(w.x, w.y, w.z, a.o_invm[g]);
        l = silent ? make_float2(a.o_lj[g].x, 0.f) : a.o_lj[g];
        fl = ghost ? 1 : 3;
        if (a.role_off_o && !ghost) nrole = a.role_off_o[g + 1] - a.role_off_o[g];
    } else {
        // dummy: far away, every dummy at its own coordinate so no two coincide
        const float d = MDX_DUMMY_BASE + MDX_DUMMY_STEP * (float)(s & 0xFFFFF);
        p = make_float4(d, d + 17.0f * (float)(s >> 20), d, 0.f);
        v = make_float4(0.f, 0.f, 0.f, 0.f);
        l = make_float2(0.f, 0.f);
    }
    a.posq[s] = p; a.vel[s] = v; a.lj[s] = l; a.ref[s] = make_float4(p.x, p.y, p.z, 0.f);
    a.path[s] = 0.f; a.dprune[s] = 0.f;
    a.force[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    a.slot_flags[s] = fl;
    if (a.role_cnt_s) {
        a.role_cnt_s[s] = nrole;
        if (t == T && lane == 0) a.role_cnt_s[(T + 1u) * MDX_TILE] = 0u;     // the scan's trailing element
    }
    // cluster bounding boxes: eight consecutive lanes
    float lo[3] = {valid ? p.x : 3.0e38f, valid ? p.y : 3.0e38f, valid ? p.z : 3.0e38f};
    float hi[3] = {valid ? p.x : -3.0e38f, valid ? p.y : -3.0e38f, valid ? p.z : -3.0e38f};
#pragma unroll
    for (int m = 1; m < 8; m <<= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], m)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], m)); }
    const unsigned long long bv = __ballot(valid), bo = __ballot((fl & 2u) != 0);
    const unsigned long long kl = __ballot(l.y != 0.f), kq = __ballot(p.w != 0.f);     // (-0.0f, the coupled atom without a well, counts as none)
    if ((lane & 7) == 0) {
        const uint32_t c = t * MDX_CL_PER_TILE + (uint32_t)(lane >> 3);
        a.cl_lo[c] = make_float4(lo[0], lo[1], lo[2], (float)__popcll((bv >> lane) & 0xFFull));
        a.cl_hi[c] = make_float4(hi[0], hi[1], hi[2], (float)__popcll((bo >> lane) & 0xFFull));   // .w: atoms this rank owns (half list)
        a.cl_kind[c] = (uint8_t)((((kl >> lane) & 0xFFull) ? 1u : 0u) | (((kq >> lane) & 0xFFull) ? 2u : 0u));
    }
}

struct FinishArgs {
    uint32_t* rb_ctl; const ListCounts* counts;
    uint32_t* tile_lpt; uint32_t lpt_grouped; WptRule rule;      // tile_lpt: tiles by list length (grouped: inside each XCD range of the pair kernel)
    const uint32_t* tile_int; uint32_t* tile_order;              // tile_order: interior tiles first, either part by length
    const unsigned long long* cursors; unsigned long long* pair_count; const uint32_t* flags;
    volatile uint32_t* host; uint32_t seq;
};
__global__ __launch_bounds__(1024) void rb_finish_kernel(FinishArgs a) {
    constexpr uint32_t NB = 8 * LPT_BUCKETS;
    __shared__ uint32_t s_hist[NB], s_cur[NB], s_scan[16];
    __shared__ uint32_t s_nint;
    const uint32_t T = a.rb_ctl[RB_T];
    if (threadIdx.x == 0) s_nint = 0u;
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t* const order = pass == 0 ? a.tile_lpt : a.tile_order;
        if (!order) continue;           // (uniform)
        uint32_t group_tiles = 0;
        if (pass == 0 && a.lpt_grouped) {      // the pair kernel's XCD ranges: ceil(workgroups / 8) workgroups of TPB tiles each (mdx_nonbonded.hip)
            const uint32_t wpt = (uint32_t)mdx_wpt_rule(a.rule, T), tpb = max(wpt, (uint32_t)MDX_NB_WAVES) / wpt;
            group_tiles = (((T + tpb - 1u) / tpb + 7u) >> 3) * tpb;
        }
        for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) s_hist[b] = 0;
        __syncthreads();
        auto bucket = [&](uint32_t t) {
            const uint32_t nch = (a.counts[t].n_masked + a.counts[t].n_plain) >> 3;
            const uint32_t major = pass == 0 ? (group_tiles ? min(t / group_tiles, 7u) : 0u) : (a.tile_int[t] ? 0u : 1u);
            return major * LPT_BUCKETS + LPT_BUCKETS - 1u - min(nch, LPT_BUCKETS - 1u);   // bucket 0 = the longest lists
        };
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) atomicAdd(&s_hist[bucket(t)], 1u);
        __syncthreads();
        {   // exclusive prefix over the NB = 1032 buckets, two per thread (one thread walking them took ~20 us of LDS round trips)
            const uint32_t b0 = 2u * threadIdx.x, h0 = b0 < NB ? s_hist[b0] : 0u, h1 = b0 + 1u < NB ? s_hist[b0 + 1u] : 0u;
            uint32_t tot;
            const uint32_t ex = block_exclusive_scan_1024(h0 + h1, s_scan, &tot);
            if (b0 < NB) s_cur[b0] = ex;
            if (b0 + 1u < NB) s_cur[b0 + 1u] = ex + h0;
            __syncthreads();
            if (pass == 1 && threadIdx.x == 0) s_nint = s_cur[LPT_BUCKETS];      // interior tiles = everything in front of major 1
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) order[atomicAdd(&s_cur[bucket(t)], 1u)] = t;
        __syncthreads();
    }
    if (threadIdx.x < 64) {      // one wave: totals of the region cursors and of the spread statistics, then the read-back
        const unsigned long long c = a.cursors[(size_t)threadIdx.x * 16];
        uint32_t e = (uint32_t)c, m = (uint32_t)(c >> 32);
        unsigned long long pa = a.pair_count[(size_t)threadIdx.x * PC_STRIDE], pb = a.pair_count[(size_t)threadIdx.x * PC_STRIDE + 1];
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) { e += __shfl_xor(e, k); m += __shfl_xor(m, k); pa += __shfl_xor(pa, k); pb += __shfl_xor(pb, k); }
        if (threadIdx.x == 0) {
            a.pair_count[PC_TOTAL] = pa; a.pair_count[PC_TOTAL + 1] = pb;
            a.rb_ctl[RB_N_INT] = s_nint;
            a.host[0] = T; a.host[1] = a.rb_ctl[RB_T_OWN]; a.host[2] = a.rb_ctl[RB_NONFINITE]; a.host[3] = s_nint;
            for (int k = 0; k < 4; ++k) a.host[4 + k] = a.flags[k];
            a.host[8] = e; a.host[9] = m;
            a.host[10] = (uint32_t)pa; a.host[11] = (uint32_t)(pa >> 32); a.host[12] = (uint32_t)pb; a.host[13] = (uint32_t)(pb >> 32);
            a.rb_ctl[RB_NONFINITE] = 0u;
            __threadfence_system();
            a.host[31] = a.seq;      // the host spins on this word
        }
    }
}


static int spin_on_readback(mdx_handle* h, uint32_t seq) {
    static const bool spin_ok = [] { const char* e = std::getenv("MDX_CHUNK_SPIN"); return !(e && e[0] == '0'); }();
    if (!spin_ok) { HIP_TRY(hipStreamSynchronize(h->stream)); return MDX_OK; }
    volatile uint32_t* seq_word = (volatile uint32_t*)h->h_rb + 31;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *seq_word != seq; ++spins) {
        if ((spins & 0xFFFu) == 0xFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            HIP_TRY(hipStreamSynchronize(h->stream));       // (an error on the stream, or a very long queue: wait the ordinary way)
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return MDX_OK;
}

struct RebuildResult { uint32_t T = 0, E = 0, MC = 0, n_interior = 0; unsigned long long npairs = 0; bool ordered = false, classified = false, inner_built = false; };

// which tile orders the rebuild produces (shared by both chains)
struct TileOrderPlan { bool lpt, grouped, split; };
static TileOrderPlan tile_order_plan(const mdx_handle* h, uint32_t T) {
    // longest lists first where a launch is only a few rounds of waves (MDX_TILE_LPT=0 / 1 forces it off / on for A/B)
    static const int lpt_env = [] { const char* e = std::getenv("MDX_TILE_LPT"); return e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : -1; }();
    // default: every decomposed handle (owned bricks and halo shells: its lists differ by an order of magnitude), and
    // single-device systems below the size at which a contiguous eighth of the tiles per XCD starts to matter
    // (later in round 3: single-device systems of every size from 512 tiles on, ordered by length inside each XCD's contiguous
    // range - water1M 1832 -> 1853 steps/s, against 1843-1849 with the round-robin order; MDX_TILE_LPT=1 / 2 force either)
    const bool decomposed = h->dd || h->n_local != h->N;
    TileOrderPlan p;
    p.lpt = (lpt_env >= 0 ? lpt_env >= 1 : (decomposed || T >= 512u)) && mdx_nb_variant(h) >= 2;
    p.grouped = lpt_env >= 0 ? lpt_env == 2 : !decomposed;
    p.split = h->want_tile_split && mdx_nb_variant(h) >= 2;
    return p;
}

// *fell_back = true: nothing was decided here (the list outgrew its arrays, or a tile its LDS buffers): the caller runs the
// unfused chain, which sizes the arrays afresh.  The slot-space state is complete either way.
static int rebuild_fast(mdx_handle* h, RebuildResult* res, bool* fell_back) {
    DeviceState& d = h->d;
    hipStream_t st = h->stream;
    const uint32_t N = h->n_local;
    const GridParams g = h->grid;
    *fell_back = false;
    if (!d.rb_ctl) { ALLOC(d.rb_ctl, RB_WORDS); HIP_TRY(hipMemsetAsync(d.rb_ctl, 0, sizeof(uint32_t) * RB_WORDS, st)); }
    if (!d.pair_count) ALLOC(d.pair_count, PC_TOTAL + 2);
    if (!d.list_cursors) ALLOC(d.list_cursors, LB_REGIONS * 32 + 2);
    if (!h->h_rb) { HIP_TRY(hipHostMalloc((void**)&h->h_rb, sizeof(uint32_t) * 32, hipHostMallocDefault)); h->h_rb[31] = 0u; }
    if (!h->cell_count_clean) { HIP_TRY(hipMemsetAsync(d.cell_count, 0, sizeof(uint32_t) * ((size_t)h->ncells + 1), st)); h->cell_count_clean = true; }
    if (!h->slot_of_clean) { HIP_TRY(hipMemsetAsync(d.slot_of, 0xFF, sizeof(uint32_t) * (size_t)h->N, st)); h->slot_of_clean = true; }
    const uint32_t T_bound = N / MDX_TILE + h->ncol + 1;        // sum over columns of ceil(n / 64) <= N / 64 + ncol
    if (T_bound + 1 > h->cap_tiles) { *fell_back = true; return MDX_OK; }      // (cannot happen behind setup_grid; if it does, the unfused chain sizes its arrays itself)
    const TileOrderPlan plan = tile_order_plan(h, T_bound);
    const bool prune = !std::isinf(h->r_list) && mdx_nb_variant(h) >= 2;
    static const bool exact_prune = [] { const char* e = std::getenv("MDX_EXACT_PRUNE"); return !(e && e[0] == '0'); }();
    // interior / boundary tiles come out of the pruning pass when owned and ghost atoms sit in separate column populations
    const bool classify_here = plan.split && plan.lpt && g.npop > 1 && prune && exact_prune;
    if (plan.lpt && (!d.tile_lpt || d.cap_tile_lpt < T_bound)) { d.cap_tile_lpt = h->cap_tiles + 1; ALLOC(d.tile_lpt, d.cap_tile_lpt); }
    if (classify_here && (!d.tile_bnd || h->cap_tile_split < T_bound + 1)) {
        h->cap_tile_split = h->cap_tiles + 1;
        ALLOC(d.tile_bnd, h->cap_tile_split); ALLOC(d.tile_scan, h->cap_tile_split); ALLOC(d.tile_order, h->cap_tile_split);
    }

    // ---- prep: caller-order positions (wrapped) and velocities, cells ----
    if (h->in_slot_space)
        hipLaunchKernelGGL(rb_prep_kernel<true>, dim3(div_up(N, 256)), dim3(256), 0, st, N, g, d.gid, d.slot_of, d.posq, d.vel, d.pos_orig,
                           d.vel_orig, d.lflag, d.cell_of, d.sorted_orig, d.cell_count, d.rb_ctl);
    else
        hipLaunchKernelGGL(rb_prep_kernel<false>, dim3(div_up(N, 256)), dim3(256), 0, st, N, g, d.gid, d.slot_of, d.posq, d.vel, d.pos_orig,
                           d.vel_orig, d.lflag, d.cell_of, d.sorted_orig, d.cell_count, d.rb_ctl);
    h->in_slot_space = false;
    {
        GridScanArgs ga{};
        ga.ncells = h->ncells; ga.ncol = h->ncol; ga.nzb = g.nzb;
        ga.first_ghost_col = g.npop > 1 ? (uint32_t)(g.ncx * g.ncy) : 0xFFFFFFFFu;
        ga.cell_count = d.cell_count; ga.cell_start = d.cell_start; ga.tile_start = d.tile_start; ga.tile_col = d.tile_col;
        ga.rb_ctl = d.rb_ctl; ga.flags = d.flags_dev; ga.pair_count = d.pair_count; ga.list_cursors = d.list_cursors;
        MDX_TRY(scan_chain_of(h, &ga.chain));
        hipLaunchKernelGGL(rb_gridscan_kernel, dim3(div_up(h->ncells + 1, SCAN1_WINDOW)), dim3(SCAN1_THREADS), 0, st, ga);
    }
    hipLaunchKernelGGL(rb_scatter_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, N, d.cell_of, d.sorted_orig, d.cell_start, d.sorted_tmp);
    hipLaunchKernelGGL(cell_rank_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, N, d.cell_of, d.cell_start, d.pos_orig, d.sorted_tmp, d.sorted_orig);
    {
        AssignArgs aa{};
        aa.rb_ctl = d.rb_ctl; aa.nzb = g.nzb; aa.tile_col = d.tile_col; aa.tile_start = d.tile_start; aa.cell_start = d.cell_start;
        aa.sorted_orig = d.sorted_orig; aa.pos_orig = d.pos_orig; aa.vel_orig = d.vel_orig; aa.gid = d.gid; aa.lflag = d.lflag;
        aa.o_qs = d.o_qs; aa.o_lj = d.o_lj; aa.o_invm = d.o_invm; aa.orig_of = d.orig_of; aa.slot_of = d.slot_of;
        aa.cl_kind = d.cl_kind; aa.kind_split = h->kind_split ? 1 : 0;
        aa.posq = d.posq; aa.lj = d.lj; aa.vel = d.vel; aa.ref = d.ref; aa.force = d.force; aa.slot_flags = d.slot_flags; aa.path = d.path; aa.dprune = d.dprune;
        aa.cl_lo = d.cl_lo; aa.cl_hi = d.cl_hi;
        aa.role_off_o = h->n_roles ? d.role_off_o : nullptr; aa.role_cnt_s = h->n_roles ? d.role_cnt_s : nullptr;
        hipLaunchKernelGGL(rb_assign_kernel, dim3(div_up(T_bound + 1, 4)), dim3(256), 0, st, aa);
    }
    h->in_slot_space = true;
    const uint32_t S_bound = (T_bound + 1) * MDX_TILE;
    if (h->n_roles) MDX_TRY(scan_u32_dev_n(h, d.role_cnt_s, d.role_off_s, S_bound + 1, d.scan_tmp, d.rb_ctl + RB_SCAN_N));

    // ---- pair list: single pass, then the exact pruning ----
    ListArgs a{};
    a.T = T_bound; a.T_dev = d.rb_ctl + RB_T; a.g = g;
    a.r_build = std::isinf(h->r_list) ? h->r_list : h->r_list * (1.0f + 1e-5f) + 1e-4f;
    a.tile_col = d.tile_col; a.tile_start = d.tile_start; a.cl_lo = d.cl_lo; a.cl_hi = d.cl_hi;
    a.orig_of = d.orig_of; a.slot_of = d.slot_of; a.gid = d.gid; a.slot_flags = d.slot_flags;
    a.excl_off = d.excl_off; a.excl_idx = d.excl_idx;
    a.counts = d.list_counts; a.entry_cnt = d.entry_cnt; a.mchunk_cnt = d.mchunk_cnt;
    a.entry_off = d.entry_off; a.mchunk_off = d.mchunk_off; a.entries = d.entries; a.masks = d.masks;
    a.err = d.flags_dev; a.null_cluster = T_bound * MDX_CL_PER_TILE;
    a.mask_layout = mdx_nb_variant(h) >= 2 ? 2 : 1;
    a.half = mdx_nb_half(h) ? 1 : 0;
    a.cell_start = d.cell_start; a.posq = d.posq; a.pair_count = d.pair_count;
    a.cursors = reinterpret_cast<unsigned long long*>(d.list_cursors);
    a.n_regions = 1;
    while (a.n_regions < (uint32_t)LB_REGIONS && a.n_regions * 128u <= T_bound) a.n_regions <<= 1;      // >= 64 tiles per region
    a.region_cap_e = (uint32_t)(std::min<uint64_t>(h->cap_entries, 0xFFFFFFFFull) / a.n_regions) & ~7u;
    a.region_cap_m = h->cap_mchunks / a.n_regions;
    a.list_grid = (div_up(T_bound, LB_WAVES) + 7u) & ~7u;
    uint32_t rf_blocks = 0;
    if (h->n_roles) {
        a.rf_S = S_bound; a.rf_role_off_o = d.role_off_o; a.rf_rec_o = d.role_rec_o; a.rf_role_off_s = d.role_off_s;
        a.rf_rec_s = d.role_rec_s; a.rf_lflag = d.lflag;
        rf_blocks = div_up(S_bound, LB_WAVES * 64);
    }
    if (g.npop > 1) hipLaunchKernelGGL((build_list_kernel<LB_SINGLE, LB_PLAIN_DD>), dim3(a.list_grid + rf_blocks), dim3(LB_WAVES * 64), 0, st, a);
    else hipLaunchKernelGGL(build_list_kernel<LB_SINGLE>, dim3(a.list_grid + rf_blocks), dim3(LB_WAVES * 64), 0, st, a);
    // Dual list: the inner list of the one-wave-per-tile class comes out of the exact pruning pass (prune_list_kernel<true>), so
    // that the force call behind this rebuild walks it instead of being a pruning pass itself.  Single-device handles, the
    // conditions under which mdx_rebuild switches the dual list on (below), MDX_REBUILD_INNER=0: A/B knob.
    // Measured (round 6, profiles/r06_rebuild_inner_ab.txt): the pass costs +96 us per rebuild at 1 M atoms (quick accept off, a second
    // ballot per surviving cluster pair) and saves the 151 us a pruning pass costs over an inner-list walk: 1936.6 -> 1946.6 steps/s on
    // the flexible headline box.  At the reference's default operating point (rigid OPC, dt 2 fs: a rebuild every 8 steps, the
    // pruning pass only 20 % dearer than the walk, kind filter in the pass) it LOSES: 864 -> 853 steps/s.  So: flexible systems
    // only; MDX_REBUILD_INNER=0 never, =2 always (A/B).
    static const int inner_mode = [] { const char* e = std::getenv("MDX_REBUILD_INNER"); return e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 1; }();
    const bool inner_env = inner_mode == 2 || (inner_mode == 1 && !mdx_has_constraints(h) && h->n_vsites == 0);
    const float inner_skin_next = h->inner_skin_auto > 0.f ? h->inner_skin_auto : c_inner_skin(h->cfg);
    const bool dual_next = !h->dual_auto_off && mdx_nb_half(h) && prune && inner_skin_next > 0.f && inner_skin_next < h->cfg.skin &&
                           (h->n_vsites == 0 || h->vsites_convex);
    const bool inner_here = inner_env && dual_next && exact_prune && !h->dd && h->n_local == h->N && g.npop == 1 && !h->alch_on &&
                            d.entries_in && d.inner_nch;
    bool inner_launched = false;
    if (prune && exact_prune) {
        const float rb = a.r_build;
        const float shx = h->per[0] ? h->box_hi[0] - h->box_lo[0] : 0.f, shy = h->per[1] ? h->box_hi[1] - h->box_lo[1] : 0.f,
                    shz = h->per[2] ? h->box_hi[2] - h->box_lo[2] : 0.f;
        uint32_t* const tint = classify_here ? d.tile_bnd : nullptr;
        const uint8_t* const kinds = h->kind_split ? d.cl_kind : nullptr;     // cluster pairs without a common interaction kind leave the list
        // One wave per tile is a launch as long as its longest list: one rank of 8 of the 1 M-atom box - 3.3 k tiles, the owned ones
        // with ~150 chunks - took 233 us where the whole box, five times the work, takes 256.  Eight waves per tile up to
        // MDX_PRUNE_MW_BELOW tiles (the unfused chain keeps its 2048)
        static const uint32_t mw_below = [] { const char* e = std::getenv("MDX_PRUNE_MW_BELOW"); return e ? (uint32_t)std::atoi(e) : 12000u; }();
        if (T_bound < mw_below)
            hipLaunchKernelGGL(prune_list_mw_kernel<8>, dim3(T_bound), dim3(512), 0, st, T_bound, rb * rb, shx, shy, shz, d.posq, d.list_counts,
                               d.entry_off, d.entries, a.null_cluster, d.pair_count, (const uint32_t*)d.rb_ctl, tint, kinds);
        else if (inner_here) {
            PruneInner pin{};
            pin.entries_in = d.entries_in; pin.inner_nch = d.inner_nch; pin.rule = mdx_wpt_rule_of(h);
            const float rin = std::max(std::isfinite(h->cfg.lj_cutoff) && h->cfg.lj_cutoff > 0.f ? h->cfg.lj_cutoff : 0.f,
                                       std::isfinite(h->cfg.coulomb_cutoff) && h->cfg.coulomb_cutoff > 0.f ? h->cfg.coulomb_cutoff : 0.f) + inner_skin_next;
            pin.rin2 = rin * rin;      // (= NbArgs::rin2 of the pair kernel's pruning pass, mdx_nonbonded.hip)
            hipLaunchKernelGGL(prune_list_kernel<true>, dim3((div_up(T_bound, 4) + 7u) & ~7u), dim3(256), 0, st, T_bound, rb * rb, shx, shy, shz,
                               d.posq, d.list_counts, d.entry_off, d.entries, a.null_cluster, d.pair_count, (const uint32_t*)d.rb_ctl, tint, kinds, pin);
            inner_launched = true;
        } else
            hipLaunchKernelGGL(prune_list_kernel<false>, dim3((div_up(T_bound, 4) + 7u) & ~7u), dim3(256), 0, st, T_bound, rb * rb, shx, shy, shz,
                               d.posq, d.list_counts, d.entry_off, d.entries, a.null_cluster, d.pair_count, (const uint32_t*)d.rb_ctl, tint, kinds,
                               PruneInner{});
    }
    MDX_TRY(mdx_remap_constraints(h));
    // ---- finish: tile orders, totals, the one read-back ----
    const uint32_t seq = ++h->rb_seq;
    {
        FinishArgs fa{};
        fa.rb_ctl = d.rb_ctl; fa.counts = d.list_counts;
        fa.tile_lpt = plan.lpt ? d.tile_lpt : nullptr; fa.lpt_grouped = plan.grouped ? 1u : 0u; fa.rule = mdx_wpt_rule_of(h);
        fa.tile_int = classify_here ? d.tile_bnd : nullptr; fa.tile_order = classify_here ? d.tile_order : nullptr;
        fa.cursors = reinterpret_cast<const unsigned long long*>(d.list_cursors); fa.pair_count = d.pair_count; fa.flags = d.flags_dev;
        fa.host = (volatile uint32_t*)h->h_rb; fa.seq = seq;
        hipLaunchKernelGGL(rb_finish_kernel, dim3(1), dim3(1024), 0, st, fa);
    }
    HIP_TRY(hipGetLastError());
    MDX_TRY(spin_on_readback(h, seq));
    const uint32_t T = h->h_rb[0];
    uint32_t flags[4];
    for (int k = 0; k < 4; ++k) flags[k] = h->h_rb[4 + k];
    if (h->h_rb[2]) { h->cell_count_clean = false; mdx_set_error("non-finite position at neighbour rebuild"); return MDX_ENAN; }
    if (T > T_bound) { mdx_set_error("internal: tile count beyond its bound"); return MDX_EDEVICE; }
    h->T = T; h->S = (T + 1) * MDX_TILE;
    if (flags[0] & 1u) { mdx_set_error("exclusion table overflow while building the pair list"); return MDX_EPARAM; }
    if (flags[0] & (16u | 32u)) {      // a tile beyond the LDS buffers, or a list that outgrew the arrays: the unfused chain sizes them afresh
        if (std::getenv("MDX_LIST_DEBUG")) fprintf(stderr, "[mdx] fused list rebuild fell back: flags %u, T %u, cap_e %llu cap_m %u regions %u\n",
                                                   flags[0], T, (unsigned long long)h->cap_entries, h->cap_mchunks, a.n_regions);
        *fell_back = true;
        return MDX_OK;
    }
    if (flags[0] & 8u) { mdx_set_error("a constrained / virtual-site atom is missing from the local atom set"); return MDX_EPARAM; }
    if (flags[0] & 2u) { mdx_set_error("exclusion table overflow while building the pair list"); return MDX_EPARAM; }
    if (flags[0] & 4u) { mdx_set_error("a bonded partner of an owned atom is missing from the local atom set (halo too thin)"); return MDX_EPARAM; }
    res->T = T; res->E = h->h_rb[8]; res->MC = h->h_rb[9];
    const unsigned long long built = (unsigned long long)h->h_rb[10] | ((unsigned long long)h->h_rb[11] << 32);
    const unsigned long long kept = (unsigned long long)h->h_rb[12] | ((unsigned long long)h->h_rb[13] << 32);
    res->npairs = (prune && exact_prune) ? kept : built;
    h->E = res->E; h->MC = res->MC;
    res->ordered = true;
    res->inner_built = inner_launched && mdx_wpt_rule(mdx_wpt_rule_of(h), T) == 1;      // (the kernel applied the same rule to the same T)
    h->force_zeroed = true;      // rb_assign_kernel cleared every slot's force: the pair launch behind this rebuild needs no fill of its own
    h->tile_lpt_on = plan.lpt; h->tile_lpt_grouped = plan.grouped;
    h->tile_split = false; h->n_interior = 0;
    if (classify_here) { h->tile_split = true; h->n_interior = h->h_rb[3]; res->classified = true; }
    return MDX_OK;
}


int mdx_rebuild(mdx_handle* h) {
    MdxRange range_rebuild("mdx list rebuild");
    h->vsites_fresh = false;      // (the next force call constructs the virtual sites itself: one launch per rebuild)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profile) {      // (events come from the handle's pool and go back to it: creating and destroying two per rebuild was host time in front of the first kernel)
        for (hipEvent_t* e : {&e0, &e1}) {
            if (!h->ev_pool.empty()) { *e = h->ev_pool.back(); h->ev_pool.pop_back(); }
            else HIP_TRY(hipEventCreate(e));
        }
        HIP_TRY(hipEventRecord(e0, h->stream));
    }
    DeviceState& d = h->d;
    // Every rebuild but a handle's first (and a few unusual ones) takes the fused chain above: MDX_REBUILD_FUSED=0 for A/B.
    static const bool fused_env = [] { const char* e = std::getenv("MDX_REBUILD_FUSED"); return !(e && e[0] == '0'); }();
    static const bool two_pass_env0 = [] { const char* e = std::getenv("MDX_LIST_TWO_PASS"); return e && e[0] == '1'; }();
    bool fast = fused_env && !two_pass_env0 && d.entries && d.masks && h->cap_entries && h->cap_mchunks && h->T > 0 && h->n_local > 0;
    for (int k = 0; k < 3; ++k) if (!h->per[k] && !h->have_local_bounds) fast = false;      // (a vacuum system: the grid follows the atoms' bounding box)
    if (!fast) MDX_TRY(mdx_unsort_state(h));  // dynamic state -> caller-order staging
    const bool was_in_slot_space = h->in_slot_space;
    MDX_TRY(setup_grid(h));
    if (fast && (size_t)h->ncells + 1 > ((size_t)1 << 21)) { fast = false; MDX_TRY(mdx_unsort_state(h)); }   // (the grid scan is a chain of at most 32 windows)
    if (fast && was_in_slot_space && !h->in_slot_space) fast = false;      // (setup_grid had to grow the slot-space arrays and moved the state to the staging)
    const uint32_t N = h->n_local;   // atoms simulated here (all of them on a single GPU)
    const GridParams g = h->grid;
    hipStream_t st = h->stream;
    RebuildResult res;
    bool fast_done = false;
    if (fast) {
        bool fell_back = false;
        MDX_TRY(rebuild_fast(h, &res, &fell_back));
        fast_done = !fell_back;
        if (fell_back) { MDX_TRY(mdx_unsort_state(h)); h->stats.rebuild_fallbacks++; }
    }
    uint32_t E = res.E, MC = res.MC;
    unsigned long long npairs = res.npairs;
    const bool prune = !std::isinf(h->r_list) && mdx_nb_variant(h) >= 2;   // the whole-tile kernel ignores imask
    if (!fast_done) {
    h->cell_count_clean = false;     // (this chain leaves the histogram behind)
    {   // slot_of = invalid, cell counters and cursors = 0, error bits = 0: one launch instead of four fills
        const uint32_t n_max = std::max<uint32_t>(h->N, h->ncells + 1);
        hipLaunchKernelGGL(rebuild_clear_kernel, dim3(div_up(n_max, 256)), dim3(256), 0, st, d.slot_of, h->N, d.cell_count,
                           d.cell_cursor, h->ncells + 1, d.flags_dev);
    }
    hipLaunchKernelGGL(bin_atoms_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, d.pos_orig, N, g, d.lflag, d.cell_of,
                       d.cell_count, d.flags_dev + 1);
    MDX_TRY(mdx_exclusive_scan_u32(h, d.cell_count, d.cell_start, h->ncells + 1));
    hipLaunchKernelGGL(column_tiles_kernel, dim3(div_up(h->ncol + 1, 256)), dim3(256), 0, st, d.cell_start,
                       h->ncol, g.nzb, d.col_tiles);
    MDX_TRY(mdx_exclusive_scan_u32(h, d.col_tiles, d.tile_start, h->ncol + 1));
    hipLaunchKernelGGL(scatter_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, d.cell_of, d.cell_start,
                       d.cell_cursor, d.sorted_tmp, N);
    hipLaunchKernelGGL(cell_rank_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, N, d.cell_of, d.cell_start,
                       d.pos_orig, d.sorted_tmp, d.sorted_orig);
    h->slot_of_clean = true;
    uint32_t T = 0, flags[4] = {0, 0, 0, 0};
    MDX_TRY(readback(h, RbSrc{{d.tile_start + h->ncol, d.flags_dev, nullptr, nullptr}, {1, 4, 0, 0}}));
    T = h->h_rb[0];
    for (int k = 0; k < 4; ++k) flags[k] = h->h_rb[1 + k];
    if (flags[1]) { mdx_set_error("non-finite position at neighbour rebuild"); return MDX_ENAN; }
    if (T + 1 > h->cap_tiles) { mdx_set_error("internal: tile capacity exceeded"); return MDX_EDEVICE; }
    h->T = T;
    h->S = (T + 1) * MDX_TILE;
    const uint32_t S = h->S, NC = (T + 1) * MDX_CL_PER_TILE;

    hipLaunchKernelGGL(tile_col_kernel, dim3(div_up(h->ncol, 256)), dim3(256), 0, st, d.tile_start, h->ncol,
                       d.tile_col);
    hipLaunchKernelGGL(assign_tiles_kernel, dim3(div_up(T + 1, 4)), dim3(256), 0, st, T, g.nzb, d.tile_col,
                       d.tile_start, d.cell_start, d.sorted_orig, d.pos_orig, d.gid, d.orig_of, d.slot_of, h->kind_split ? 1 : 0,
                       d.lflag, d.o_qs, d.o_lj);
    hipLaunchKernelGGL(gather_slots_kernel, dim3(div_up(S, 256)), dim3(256), 0, st, S, d.orig_of, d.gid, d.lflag,
                       d.pos_orig, d.vel_orig, d.o_qs, d.o_lj, d.o_invm, d.posq, d.lj, d.vel, d.ref, d.force,
                       d.slot_flags, d.path, d.dprune);
    hipLaunchKernelGGL(cluster_bbox_kernel, dim3(div_up(NC, 256)), dim3(256), 0, st, NC, d.orig_of, d.posq,
                       d.slot_flags, d.cl_lo, d.cl_hi, d.lj, d.cl_kind);
    h->in_slot_space = true;

    // ---- pair list: count, scan, fill ----
    ListArgs a{};
    a.T = T; a.g = g;
    a.r_build = std::isinf(h->r_list) ? h->r_list : h->r_list * (1.0f + 1e-5f) + 1e-4f;
    a.tile_col = d.tile_col; a.tile_start = d.tile_start; a.cl_lo = d.cl_lo; a.cl_hi = d.cl_hi;
    a.orig_of = d.orig_of; a.slot_of = d.slot_of; a.gid = d.gid; a.slot_flags = d.slot_flags;
    a.excl_off = d.excl_off; a.excl_idx = d.excl_idx;
    a.counts = d.list_counts; a.entry_cnt = d.entry_cnt; a.mchunk_cnt = d.mchunk_cnt;
    a.entry_off = d.entry_off; a.mchunk_off = d.mchunk_off; a.entries = d.entries; a.masks = d.masks;
    a.err = d.flags_dev; a.null_cluster = T * MDX_CL_PER_TILE;
    a.mask_layout = mdx_nb_variant(h) >= 2 ? 2 : 1;
    a.half = mdx_nb_half(h) ? 1 : 0;
    a.cell_start = d.cell_start;
    a.posq = d.posq;
    if (!d.pair_count) ALLOC(d.pair_count, PC_TOTAL + 2);
    a.pair_count = d.pair_count;
    E = 0; MC = 0; npairs = 0;
    // Single pass when the arrays of the previous build are there to be reused (every rebuild but the first): one
    // search of every tile's neighbourhood instead of count + scan + fill.  MDX_LIST_TWO_PASS=1 keeps the two passes.
    static const bool two_pass_env = [] { const char* e = std::getenv("MDX_LIST_TWO_PASS"); return e && e[0] == '1'; }();
    // The single pass is launched and NOT waited for: the exact pruning, the bonded role lists and the constraint remap
    // queue up behind it, and its cursors and overflow bits are read at the one synchronisation at the end.  A tile that
    // did not fit has an empty list (counts 0) by then, so what ran behind it was safe; count + fill then builds afresh.
    bool speculative = false, roles_done = false;
    if (d.entries && d.masks && h->cap_entries && h->cap_mchunks && T && !two_pass_env) {
        if (!d.list_cursors) ALLOC(d.list_cursors, LB_REGIONS * 32 + 2);     // 64 regions x one 128-B line, + the two totals
        HIP_TRY(hipMemsetAsync(d.pair_count, 0, sizeof(unsigned long long) * (PC_TOTAL + 2), st));
        HIP_TRY(hipMemsetAsync(d.list_cursors, 0, sizeof(uint32_t) * (LB_REGIONS * 32 + 2), st));
        a.cursors = reinterpret_cast<unsigned long long*>(d.list_cursors);
        a.n_regions = 1;
        while (a.n_regions < (uint32_t)LB_REGIONS && a.n_regions * 128u <= T) a.n_regions <<= 1;      // >= 64 tiles per region
        a.region_cap_e = (uint32_t)(std::min<uint64_t>(h->cap_entries, 0xFFFFFFFFull) / a.n_regions) & ~7u;
        a.region_cap_m = h->cap_mchunks / a.n_regions;
        a.list_grid = (div_up(T, LB_WAVES) + 7u) & ~7u;
        uint32_t rf_blocks = 0;
        if (h->n_roles) {      // bonded role lists into slot space: count + scan here, the fill as extra workgroups of the list build
            hipLaunchKernelGGL(role_count_kernel, dim3(div_up(S + 1, 256)), dim3(256), 0, st, S, d.orig_of, d.gid,
                               d.lflag, d.role_off_o, d.role_cnt_s);
            MDX_TRY(mdx_exclusive_scan_u32(h, d.role_cnt_s, d.role_off_s, S + 1));
            a.rf_S = S; a.rf_role_off_o = d.role_off_o; a.rf_rec_o = d.role_rec_o; a.rf_role_off_s = d.role_off_s;
            a.rf_rec_s = d.role_rec_s; a.rf_lflag = d.lflag;
            rf_blocks = div_up(S, LB_WAVES * 64);
            roles_done = true;
        }
        if (g.npop > 1) hipLaunchKernelGGL((build_list_kernel<LB_SINGLE, LB_PLAIN_DD>), dim3(a.list_grid + rf_blocks), dim3(LB_WAVES * 64), 0, st, a);
        else hipLaunchKernelGGL(build_list_kernel<LB_SINGLE>, dim3(a.list_grid + rf_blocks), dim3(LB_WAVES * 64), 0, st, a);
        speculative = true;
    }
    // A region of the single-pass build fills at cap / n_regions, which an imbalance between the regions reaches before the
    // totals reach cap: while a list grows (equilibration, a shrinking box, an inhomogeneous solute) every rebuild would then
    // run the speculative pass AND count + fill with the arrays unchanged.  After such an overflow the arrays are sized with
    // half as much room again as the list needs, so the overflow is paid once (round-2 advisor finding).
    bool grow_for_regions = false;
    auto two_pass = [&]() -> int {
        HIP_TRY(hipMemsetAsync(d.pair_count, 0, sizeof(unsigned long long) * (PC_TOTAL + 2), st));
        HIP_TRY(hipMemsetAsync(d.entry_cnt + T, 0, sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(d.mchunk_cnt + T, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(build_list_kernel<LB_COUNT>, dim3((div_up(T, LB_WAVES) + 7u) & ~7u), dim3(LB_WAVES * 64), 0, st, a);
        MDX_TRY(mdx_exclusive_scan_u32(h, d.entry_cnt, d.entry_off, T + 1));
        MDX_TRY(mdx_exclusive_scan_u32(h, d.mchunk_cnt, d.mchunk_off, T + 1));
        hipLaunchKernelGGL(pair_sum_kernel, dim3(1), dim3(64), 0, st, d.pair_count);
        MDX_TRY(readback(h, RbSrc{{d.flags_dev, d.entry_off + T, d.mchunk_off + T, reinterpret_cast<const uint32_t*>(d.pair_count + PC_TOTAL)}, {4, 1, 1, 2}}));
        for (int k = 0; k < 4; ++k) flags[k] = h->h_rb[k];
        E = h->h_rb[4]; MC = h->h_rb[5];
        npairs = (unsigned long long)h->h_rb[6] | ((unsigned long long)h->h_rb[7] << 32);
        if (flags[0] & 3u) { mdx_set_error("exclusion table overflow while building the pair list"); return MDX_EPARAM; }
        if (E > h->cap_entries || !d.entries || (grow_for_regions && (double)h->cap_entries < 1.5 * (double)E)) {
            h->cap_entries = (uint64_t)(E * (grow_for_regions ? 1.5 : 1.25)) + 1024;
            ALLOC(d.entries, h->cap_entries);
            ALLOC(d.entries_in, h->cap_entries);   // dual list: the pruning pass of the pair kernel fills it
            a.entries = d.entries;
        }
        if (MC > h->cap_mchunks || !d.masks || (grow_for_regions && (double)h->cap_mchunks < 1.5 * (double)MC)) {
            h->cap_mchunks = (uint32_t)(MC * (grow_for_regions ? 1.5 : 1.25)) + 64;
            ALLOC(d.masks, (size_t)h->cap_mchunks * 64);
            a.masks = d.masks;
        }
        h->E = E; h->MC = MC;
        hipLaunchKernelGGL(build_list_kernel<LB_FILL>, dim3((div_up(T, LB_WAVES) + 7u) & ~7u), dim3(LB_WAVES * 64), 0, st, a);
        return MDX_OK;
    };
    if (!speculative) MDX_TRY(two_pass());
    static const bool exact_prune = [] { const char* e = std::getenv("MDX_EXACT_PRUNE"); return !(e && e[0] == '0'); }();   // A/B knob
    auto launch_prune = [&]() {
        if (!(prune && T && exact_prune)) return;
        const uint8_t* const kinds = h->kind_split ? d.cl_kind : nullptr;
        const float rb = a.r_build;
        const float shx = h->per[0] ? h->box_hi[0] - h->box_lo[0] : 0.f, shy = h->per[1] ? h->box_hi[1] - h->box_lo[1] : 0.f,
                    shz = h->per[2] ? h->box_hi[2] - h->box_lo[2] : 0.f;
        if (T < 2048u)          // a few hundred tiles: eight waves per tile (one wave per tile is a 1/3-empty chip waiting for its longest list)
            hipLaunchKernelGGL(prune_list_mw_kernel<8>, dim3(T), dim3(512), 0, st, T, rb * rb, shx, shy, shz, d.posq, d.list_counts,
                               d.entry_off, d.entries, a.null_cluster, d.pair_count, (const uint32_t*)nullptr, (uint32_t*)nullptr, kinds);
        else
            hipLaunchKernelGGL(prune_list_kernel<false>, dim3((div_up(T, 4) + 7u) & ~7u), dim3(256), 0, st, T, rb * rb, shx, shy, shz,
                               d.posq, d.list_counts, d.entry_off, d.entries, a.null_cluster, d.pair_count, (const uint32_t*)nullptr, (uint32_t*)nullptr, kinds,
                               PruneInner{});
    };
    launch_prune();

    // ---- bonded role lists into slot space ----
    if (h->n_roles && !roles_done) {
        hipLaunchKernelGGL(role_count_kernel, dim3(div_up(S + 1, 256)), dim3(256), 0, st, S, d.orig_of, d.gid,
                           d.lflag, d.role_off_o, d.role_cnt_s);
        MDX_TRY(mdx_exclusive_scan_u32(h, d.role_cnt_s, d.role_off_s, S + 1));
        hipLaunchKernelGGL(role_fill_kernel, dim3(div_up(S, 256)), dim3(256), 0, st, S, d.orig_of, d.gid, d.lflag,
                           d.slot_of, d.role_off_o, d.role_rec_o, d.role_off_s, d.role_rec_s, d.flags_dev);
    }
    MDX_TRY(mdx_remap_constraints(h));
    HIP_TRY(hipGetLastError());
    unsigned long long npairs_pruned = 0;
    auto read_counts = [&]() -> int {
        if (speculative) hipLaunchKernelGGL(cursor_sum_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned long long*>(d.list_cursors),
                                            d.list_cursors + LB_REGIONS * 32);
        hipLaunchKernelGGL(pair_sum_kernel, dim3(1), dim3(64), 0, st, d.pair_count);
        MDX_TRY(readback(h, RbSrc{{d.flags_dev, d.list_cursors ? d.list_cursors + LB_REGIONS * 32 : nullptr,
                                   reinterpret_cast<const uint32_t*>(d.pair_count + PC_TOTAL), nullptr}, {4, 2, 4, 0}}));
        for (int k = 0; k < 4; ++k) flags[k] = h->h_rb[k];
        if (speculative) {
            E = h->h_rb[4]; MC = h->h_rb[5];
            npairs = (unsigned long long)h->h_rb[6] | ((unsigned long long)h->h_rb[7] << 32);
        }
        npairs_pruned = (unsigned long long)h->h_rb[8] | ((unsigned long long)h->h_rb[9] << 32);
        return MDX_OK;
    };
    MDX_TRY(read_counts());
    if (speculative) {
        if (flags[0] & 1u) { mdx_set_error("exclusion table overflow while building the pair list"); return MDX_EPARAM; }
        if (flags[0] & (16u | 32u)) {
            // a tile beyond the LDS buffers, or a list that outgrew the arrays: count + fill sizes them afresh
            // (the role lists and the constraint remap above do not depend on the pair list and stand)
            const uint32_t keep = flags[0] & ~(16u | 32u);
            HIP_TRY(hipMemcpyAsync(d.flags_dev, &keep, sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            speculative = false;
            grow_for_regions = (flags[0] & 32u) != 0;
            if (std::getenv("MDX_LIST_DEBUG")) fprintf(stderr, "[mdx] single-pass list build fell back: flags %u, T %u, E %u MC %u, cap_e %llu cap_m %u regions %u\n",
                                                       flags[0], T, E, MC, (unsigned long long)h->cap_entries, h->cap_mchunks, a.n_regions);
            MDX_TRY(two_pass());
            launch_prune();
            MDX_TRY(read_counts());
        } else {
            h->E = E; h->MC = MC;
        }
    }
    if (prune && exact_prune) npairs = npairs_pruned;
    if (flags[0] & 8u) { mdx_set_error("a constrained / virtual-site atom is missing from the local atom set"); return MDX_EPARAM; }
    if (flags[0] & 3u) { mdx_set_error("exclusion table overflow while building the pair list"); return MDX_EPARAM; }
    if (flags[0] & 4u) {
        mdx_set_error("a bonded partner of an owned atom is missing from the local atom set (halo too thin)");
        return MDX_EPARAM;
    }

    res.T = T;
    }   // (the unfused chain)
    const uint32_t T = h->T, S = h->S, NC = (T + 1) * MDX_CL_PER_TILE;
    {
        const TileOrderPlan plan = tile_order_plan(h, T);
        if (!res.classified) {
            h->tile_split = false; h->n_interior = 0;
            if (plan.split) MDX_TRY(mdx_classify_tiles(h, plan.lpt));
        }
        if (!res.ordered) {
            h->tile_lpt_on = false;
            if (plan.lpt) MDX_TRY(mdx_order_tiles_by_length(h, plan.grouped));
        }
    }
    h->list_valid = true;
    h->forces_valid = false;
    h->rebuild_count++; h->steps_since_rebuild = 0;
    mdx_dd_note_rebuild(h);
    // dual pair list: the step loop of a half-list run walks a rolling-pruned inner list.  What moves an atom inside
    // the step loop feeds its path accumulator: the drift pass, SHAKE, and - for the ghosts of a decomposed handle - the
    // halo unpack; a virtual site inside the triangle of its parents never moves further than they do (anything else
    // keeps the plain list)
    h->inner_skin = h->inner_skin_auto > 0.f ? h->inner_skin_auto : c_inner_skin(h->cfg);
    h->dual_on = !h->dual_auto_off && mdx_nb_half(h) && prune && h->inner_skin > 0.f && h->inner_skin < h->cfg.skin &&
                 (h->n_vsites == 0 || h->vsites_convex);
    h->prune_pending = !(res.inner_built && h->dual_on);      // (the fused chain's pruning pass has written the inner list of these positions)
    h->inner_from_rebuild = !h->prune_pending;
    if (h->inner_from_rebuild) ++h->inner_rebuilds;
    if (!d.inner_count) { ALLOC(d.inner_count, MDX_EPART + 8); HIP_TRY(hipMemsetAsync(d.inner_count, 0, sizeof(unsigned long long) * (MDX_EPART + 8), st)); }
    uint64_t nmask = (uint64_t)MC * 8;
    h->stats.n_atoms = N; h->stats.n_slots = S; h->stats.n_tiles = T; h->stats.n_clusters = NC;
    h->stats.n_list_entries = E; h->stats.n_masked_entries = nmask;
    h->stats.n_cluster_pairs = (mdx_nb_variant(h) >= 2) ? npairs : (uint64_t)E * 8;
    if (h->profile) {
        float ms = 0.f;
        HIP_TRY(hipEventRecord(e1, st)); HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        h->stats.rebuild_ms_sum += ms;
        h->ev_pool.push_back(e0); h->ev_pool.push_back(e1);
    }
    return MDX_OK;
}

// ---- tiles by decreasing list length ------------------------------------------------------------------------------------
// A launch of a few thousand tiles is 2-3 rounds of waves per SIMD; in spatial order the long lists (owned bricks, box
// interior) and the short ones (halo shells, faces of a non-periodic system) come in runs, and the launch ends when the
// last long list does.  Longest first is the classic list-scheduling order: the short lists fill the tail.  A counting
// sort over the chunk count (129 buckets), one small launch per rebuild.
// One workgroup, LDS histogram and cursors (atomics on a hundred global words would queue: a returning atomic on one address is
// served every ~35 ns, tools/ubench/grid_barrier.hip - 70 us for the fullest bucket of a 9 k-tile launch).
// is_int (may be null): the interior / boundary split of a decomposed handle - interior tiles first, each part by length.
// group_tiles > 0 (large single-device systems): the order is by length INSIDE each of the eight contiguous tile ranges the pair
// kernel hands to the eight XCDs (group_tiles tiles each), so a range stays on its XCD's L2 and still runs longest list first.
__global__ __launch_bounds__(1024) void lpt_order_kernel(uint32_t T, const ListCounts* __restrict__ counts, const uint32_t* __restrict__ is_int,
                                                         uint32_t* __restrict__ order, uint32_t group_tiles) {
    constexpr uint32_t NB = 8 * LPT_BUCKETS;
    __shared__ uint32_t s_hist[NB], s_cur[NB];
    for (uint32_t b = threadIdx.x; b < NB; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    auto bucket = [&](uint32_t t) {
        const uint32_t nch = (counts[t].n_masked + counts[t].n_plain) >> 3;
        const uint32_t major = group_tiles ? min(t / group_tiles, 7u) : ((is_int && !is_int[t]) ? 1u : 0u);
        return major * LPT_BUCKETS + LPT_BUCKETS - 1u - min(nch, LPT_BUCKETS - 1u);   // bucket 0 = the longest lists
    };
    for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) atomicAdd(&s_hist[bucket(t)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < NB; ++b) { s_cur[b] = run; run += s_hist[b]; }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) order[atomicAdd(&s_cur[bucket(t)], 1u)] = t;
}

// ---- decomposed handle: interior / boundary tiles ---------------------------------------------------------------------
// A tile is a BOUNDARY tile when it holds a ghost atom or its (Verlet) list names a cluster that does; everything an
// interior tile's pair evaluation reads is owned by this rank, so it can run before the halo message has arrived.
__global__ __launch_bounds__(256) void tile_class_kernel(uint32_t T, const uint8_t* __restrict__ slot_flags,
                                                         const ListCounts* __restrict__ counts, const uint32_t* __restrict__ entry_off,
                                                         const uint2* __restrict__ entries, uint32_t* __restrict__ tile_bnd) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= T) return;
    const uint8_t f = slot_flags[t * MDX_TILE + lane];
    bool ghost = (f & 1u) && !(f & 2u);
    const ListCounts c = counts[t];
    const uint32_t e0 = entry_off[t], n = c.n_masked + c.n_plain;
    for (uint32_t k = lane; k < n; k += 64) {
        const uint2 e = entries[e0 + k];
        if (((e.y >> 8) & 0xFFu) == 0u) continue;
        const unsigned long long fl = *reinterpret_cast<const unsigned long long*>(slot_flags + (size_t)e.x * MDX_CLUSTER);
        // a byte with bit0 set and bit1 clear: real atom, not owned
        ghost |= ((fl & 0x0101010101010101ull) & ~((fl >> 1) & 0x0101010101010101ull)) != 0ull;
    }
    const bool any = __any(ghost);
    if (lane == 0) tile_bnd[t] = any ? 0u : 1u;     // stored as "is interior" so that the scan counts interior tiles
}
__global__ void tile_order_kernel(uint32_t T, const uint32_t* __restrict__ is_int, const uint32_t* __restrict__ scan,
                                  uint32_t* __restrict__ order) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t n_int = scan[T];
    order[is_int[t] ? scan[t] : n_int + (t - scan[t])] = t;
}

int mdx_classify_tiles(mdx_handle* h, bool by_length) {
    DeviceState& d = h->d;
    const uint32_t T = h->T;
    hipStream_t st = h->stream;
    h->tile_split = false; h->n_interior = 0;
    if (!T) return MDX_OK;
    if (!d.tile_bnd || h->cap_tile_split < T + 1) {
        h->cap_tile_split = h->cap_tiles + 1;
        ALLOC(d.tile_bnd, h->cap_tile_split); ALLOC(d.tile_scan, h->cap_tile_split); ALLOC(d.tile_order, h->cap_tile_split);
    }
    HIP_TRY(hipMemsetAsync(d.tile_bnd + T, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(tile_class_kernel, dim3(div_up(T, 4)), dim3(256), 0, st, T, d.slot_flags, d.list_counts, d.entry_off, d.entries,
                       d.tile_bnd);
    MDX_TRY(mdx_exclusive_scan_u32(h, d.tile_bnd, d.tile_scan, T + 1));
    if (by_length) hipLaunchKernelGGL(lpt_order_kernel, dim3(1), dim3(1024), 0, st, T, d.list_counts, d.tile_bnd, d.tile_order, 0u);
    else hipLaunchKernelGGL(tile_order_kernel, dim3(div_up(T, 256)), dim3(256), 0, st, T, d.tile_bnd, d.tile_scan, d.tile_order);
    uint32_t n_int = 0;
    HIP_TRY(hipMemcpyAsync(&n_int, d.tile_scan + T, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->n_interior = n_int; h->tile_split = true;
    return MDX_OK;
}

int mdx_order_tiles_by_length(mdx_handle* h, bool grouped) {
    DeviceState& d = h->d;
    const uint32_t T = h->T;
    hipStream_t st = h->stream;
    h->tile_lpt_on = false;
    if (!T) return MDX_OK;
    if (!d.tile_lpt || d.cap_tile_lpt < T) {
        d.cap_tile_lpt = h->cap_tiles + 1;
        ALLOC(d.tile_lpt, d.cap_tile_lpt);
    }
    uint32_t group_tiles = 0;
    if (grouped) {      // the pair kernel's XCD ranges: ceil(workgroups / 8) workgroups of TPB tiles each (mdx_nonbonded.hip)
        const uint32_t wpt = (uint32_t)mdx_nb_wpt_half(h, T), tpb = std::max(wpt, (uint32_t)MDX_NB_WAVES) / wpt;
        group_tiles = (((T + tpb - 1) / tpb + 7) >> 3) * tpb;
    }
    hipLaunchKernelGGL(lpt_order_kernel, dim3(1), dim3(1024), 0, st, T, d.list_counts, (const uint32_t*)nullptr, d.tile_lpt, group_tiles);
    HIP_TRY(hipGetLastError());
    h->tile_lpt_on = true; h->tile_lpt_grouped = grouped;
    return MDX_OK;
}

int mdx_extract_neighbors(mdx_handle* h, uint32_t* offsets, uint32_t* idx) {
    DeviceState& d = h->d;
    const uint32_t N = h->n_local, T = h->T;
    hipStream_t st = h->stream;
    const float rl2 = h->r_list * h->r_list;
    uint32_t *d_cnt = nullptr, *d_off = nullptr, *d_idx = nullptr, *d_cur = nullptr;
    const int half = mdx_nb_half(h) ? 1 : 0;
    HIP_TRY(hipMalloc((void**)&d_cnt, sizeof(uint32_t) * ((size_t)N + 1)));
    HIP_TRY(hipMalloc((void**)&d_off, sizeof(uint32_t) * ((size_t)N + 1)));
    HIP_TRY(hipMalloc((void**)&d_cur, sizeof(uint32_t) * ((size_t)N + 1)));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * ((size_t)N + 1), st));
    HIP_TRY(hipMemsetAsync(d_cur, 0, sizeof(uint32_t) * ((size_t)N + 1), st));
    hipLaunchKernelGGL(extract_neighbors_kernel<false>, dim3(div_up(T, 4)), dim3(256), 0, st, T, h->grid, rl2, half,
                       d.entry_off, d.list_counts, d.entries, d.orig_of, d.ref, d_cnt, d_off, d_cur, d_idx);
    int rc = mdx_exclusive_scan_u32(h, d_cnt, d_off, N + 1);
    if (rc != MDX_OK) { (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_cur); return rc; }
    std::vector<uint32_t> off(N + 1);
    HIP_TRY(hipMemcpyAsync(off.data(), d_off, sizeof(uint32_t) * ((size_t)N + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t total = off[N];
    std::vector<uint32_t> raw(total ? total : 1);
    HIP_TRY(hipMalloc((void**)&d_idx, sizeof(uint32_t) * (total ? total : 1)));
    hipLaunchKernelGGL(extract_neighbors_kernel<true>, dim3(div_up(T, 4)), dim3(256), 0, st, T, h->grid, rl2, half,
                       d.entry_off, d.list_counts, d.entries, d.orig_of, d.ref, d_cnt, d_off, d_cur, d_idx);
    HIP_TRY(hipMemcpyAsync(raw.data(), d_idx, sizeof(uint32_t) * total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_idx); (void)hipFree(d_cur);
    // rows: sort, drop duplicates (a cluster may be listed under two images in a small box)
    uint32_t w = 0;
    std::vector<uint32_t> new_off(N + 1);
    for (uint32_t i = 0; i < N; ++i) {
        new_off[i] = w;
        uint32_t* b = raw.data() + off[i];
        uint32_t* e = raw.data() + off[i + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        const uint32_t n = (uint32_t)(e - b);
        if (idx) std::copy(b, e, idx + w);
        w += n;
    }
    new_off[N] = w;
    std::copy(new_off.begin(), new_off.end(), offsets);
    return MDX_OK;
}
