// mdx_decomp.hip — spatial domain decomposition of one periodic box over the GPUs of a node, below the C ABI
// (the reference is single-device, /root/reference src/util.rs:1086, so this is new capability).
//
//   * The box is cut into px x py x pz bricks (2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2: with periodic wrap every rank of a
//     2x2x2 grid has exactly 7 peers = the 7 xGMI links of an MI355X).  Every rank holds one handle created from the
//     GLOBAL system: static per-atom data and topology are replicated (288 GB of HBM make that free), only dynamic
//     state is distributed.
//   * A rank integrates the atoms it OWNS and keeps GHOST copies of every other atom within halo = r_list + margin
//     (+ the reach of a constraint cluster) of its brick, shifted into its own frame: a cut dimension is not periodic
//     locally.  Ownership is decided per constraint cluster / virtual-site family (by the position of its first atom),
//     so SHAKE, RATTLE and the construction / force spreading of virtual sites never cross a rank boundary.
//   * Per step: drift (+ SHAKE, + virtual sites) -> pack -> ONE group of ncclSend / ncclRecv on the communication
//     stream -> unpack -> forces -> ghost forces back.  The halo is a HALF shell (default, with the half-list pair kernel):
//     a rank keeps ghosts only of atoms whose owner's brick lies in an upper direction, so a pair of atoms owned by two ranks
//     is evaluated on exactly one of them, and the forces that rank computed on its ghosts return to their owners in a second
//     send/recv group along the same segments (MDX_HALF_SHELL=0: full shell, every rank evaluates every pair of its owned
//     atoms, no force message).  The "list went stale" word of every rank rides on both messages (each peer's segment ends
//     with a flag row), so all ranks stop at the same step without a separate collective.
//   * A stale list is rebuilt LOCALLY while the owned + ghost set is still complete (no atom further than margin/2 from
//     where it was at the last repartition: one small all-reduce decides, the same way on every rank); otherwise the
//     ranks REPARTITION: every rank's owned rows are gathered everywhere (one group of sends/receives: each rank's block
//     goes straight to its 7 peers) and every rank re-derives owners, ghosts, image shifts and both halo lists from the
//     same data with the same arithmetic, so no index list is ever exchanged.
// All partition arithmetic runs in device kernels in fp32 and is bit-identical on every rank.
#include "mdx_comm.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#define FAIL(code, msg) do { mdx_set_error(msg); return (code); } while (0)
static inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

int mdx_exclusive_scan_u32_ex(mdx_handle* h, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums);   // mdx_grid.hip
int mdx_set_local_atoms_impl(mdx_handle* h, uint32_t n_local, const uint32_t* d_gid, const uint8_t* d_ghost, const float* d_pos4,
                             const float* d_vel4, const float lo[3], const float hi[3], int32_t periodic);     // mdx_api.hip

#define DD_MAX_WORLD 32

struct DdPart {
    float lo[3], len[3];
    int grid[3], coord[3];
    int rank, world;
    float halo;
};

__device__ __forceinline__ float dd_wrap1(float x, float lo, float L) {
    float t = x - floorf((x - lo) / L) * L;
    if (t < lo) t += L;
    if (t >= lo + L) t -= L;
    return t;
}
__device__ __forceinline__ int dd_owner(const DdPart& p, float x, float y, float z) {
    const float v[3] = {x, y, z};
    int c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int k = (int)floorf((v[d] - p.lo[d]) / p.len[d] * (float)p.grid[d]);
        c[d] = min(max(k, 0), p.grid[d] - 1);
    }
    return (c[0] * p.grid[1] + c[1]) * p.grid[2] + c[2];
}
// Is the (wrapped) point inside brick `c` widened by the halo, under one of the images k = -1, 0, +1 of every cut
// dimension?  code: (kx + 1) | (ky + 1) << 2 | (kz + 1) << 4 of the first image that is.
__device__ __forceinline__ bool dd_in_halo(const DdPart& p, const int c[3], float x, float y, float z, uint32_t* code) {
    const float v[3] = {x, y, z};
    uint32_t cd = 1u | (1u << 2) | (1u << 4);
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (p.grid[d] == 1) continue;
        const float blo = p.lo[d] + p.len[d] * (float)c[d] / (float)p.grid[d];
        const float bhi = p.lo[d] + p.len[d] * (float)(c[d] + 1) / (float)p.grid[d];
        const float lo_h = blo - p.halo, hi_h = bhi + p.halo;
        int kk = 2;
#pragma unroll
        for (int k = -1; k <= 1; ++k) {
            const float xs = v[d] + (float)k * p.len[d];
            if (kk == 2 && xs >= lo_h && xs < hi_h) kk = k;
        }
        if (kk == 2) ok = false;
        else cd = (cd & ~(3u << (2 * d))) | ((uint32_t)(kk + 1) << (2 * d));
    }
    *code = cd;
    return ok;
}

// Half shell: does rank (brick) `c` keep a ghost of an atom owned by brick `co`, which it sees under image `code` while the
// owner itself holds the atom under image `home` (non-zero for a cluster member that lies across the periodic seam from its
// anchor)?  Only when the owner's brick - in the frame the atom lives in - lies in an UPPER direction:
//     rel_d = co_d + (k_d - home_d) grid_d - c_d   over the cut dimensions, first non-zero component positive.
// For two atoms i (owner P, home a_i) and j (owner Q, home a_j) within range of each other the interacting image is unique,
// so k^P_j - a_i = a_j - k^Q_i, hence rel seen from Q is -rel seen from P: exactly one of the two owners keeps the other's
// atom as a ghost, the pair is evaluated once, and its force on the ghost travels back.
__device__ __forceinline__ bool dd_upper(const DdPart& p, const int c[3], const int co[3], uint32_t code, uint32_t home) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (p.grid[d] == 1) continue;
        const int k = (int)((code >> (2 * d)) & 3u) - (int)((home >> (2 * d)) & 3u);
        const int rel = co[d] + k * p.grid[d] - c[d];
        if (rel != 0) return rel > 0;
    }
    return false;
}
__device__ __forceinline__ int dd_owner_of(const DdPart& p, const float4* __restrict__ g_pos, const uint32_t* __restrict__ anchor, uint32_t g) {
    const float4 pa = g_pos[anchor[g]];
    return dd_owner(p, dd_wrap1(pa.x, p.lo[0], p.len[0]), dd_wrap1(pa.y, p.lo[1], p.len[1]), dd_wrap1(pa.z, p.lo[2], p.len[2]));
}

// one pass over the global state: owner, class here, image code, and - for owned atoms - the peers that keep a ghost.
// Classes: 0 not here, 1 owned, 2 ghost, 3 (half shell) ghost kept only because an owned atom has a bonded term with it: the
// atom-owned bonded gather needs its position, its pair interactions belong to other ranks (no charge, no LJ here).
__global__ __launch_bounds__(256) void dd_classify_kernel(uint32_t N, const float4* __restrict__ g_pos, const uint32_t* __restrict__ anchor,
                                                          DdPart p, int half_shell, const uint32_t* __restrict__ role_off,
                                                          const RoleRec* __restrict__ roles,
                                                          uint8_t* __restrict__ cls, uint8_t* __restrict__ owner,
                                                          uint8_t* __restrict__ shift_code, uint32_t* __restrict__ send_mask,
                                                          uint32_t* __restrict__ err) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const float4 pg = g_pos[g];
    const int own = dd_owner_of(p, g_pos, anchor, g);
    const float x = dd_wrap1(pg.x, p.lo[0], p.len[0]), y = dd_wrap1(pg.y, p.lo[1], p.len[1]), z = dd_wrap1(pg.z, p.lo[2], p.len[2]);
    const int co[3] = {own / (p.grid[1] * p.grid[2]), (own / p.grid[2]) % p.grid[1], own % p.grid[2]};
    uint32_t code, home = 1u | (1u << 2) | (1u << 4);
    const bool here = dd_in_halo(p, p.coord, x, y, z, &code);
    const bool mine = own == p.rank;
    if (mine && !here) atomicOr(err, 1u);      // an owned atom outside its own halo region: the halo is thinner than the cluster reach
    if (half_shell && !mine) (void)dd_in_halo(p, co, x, y, z, &home);   // the image its owner holds it under (the owner checks that it can)
    if (mine) home = code;
    uint32_t c = mine ? 1u : (here ? 2u : 0u);
    if (half_shell && c == 2u && !dd_upper(p, p.coord, co, code, home)) c = 0u;
    uint32_t mask = 0;
    if (mine) {
        for (int q = 0; q < p.world; ++q) {
            if (q == p.rank) continue;
            const int cq[3] = {q / (p.grid[1] * p.grid[2]), (q / p.grid[2]) % p.grid[1], q % p.grid[2]};
            uint32_t cd;
            if (dd_in_halo(p, cq, x, y, z, &cd) && (!half_shell || dd_upper(p, cq, co, cd, home))) mask |= 1u << q;
        }
    }
    if (half_shell && role_off) {
        // bonded partners across a rank boundary travel in BOTH directions: whoever owns one atom of a term needs the others
        for (uint32_t k = role_off[g]; k < role_off[g + 1]; ++k) {
            const RoleRec r = roles[k];
            const uint32_t kind = r.meta & 0xFu;
            const int np = kind == ROLE_DIHEDRAL ? 3 : (kind == ROLE_ANGLE ? 2 : 1);
            for (int q = 0; q < np; ++q) {
                const int po = dd_owner_of(p, g_pos, anchor, r.p[q]);
                if (mine) {
                    if (po != p.rank) {
                        const int cq[3] = {po / (p.grid[1] * p.grid[2]), (po / p.grid[2]) % p.grid[1], po % p.grid[2]};
                        uint32_t cd;
                        if (dd_in_halo(p, cq, x, y, z, &cd)) mask |= 1u << po;
                        else { atomicOr(err, 4u); if (atomicCAS(err + 3, 0u, (uint32_t)po | 0x100u) == 0u) { err[1] = g; err[2] = r.p[q]; } }   // (the receiver finds the same and fails alike)
                    }
                } else if (po == p.rank && c == 0u) {
                    if (here) c = 3u; else { atomicOr(err, 4u); if (atomicCAS(err + 3, 0u, (uint32_t)own | 0x200u) == 0u) { err[1] = g; err[2] = r.p[q]; } }   // a bonded partner beyond the halo
                }
            }
        }
    }
    // the folded drift pass keeps at most seven message rows per slot (one per peer of a 2 x 2 x 2 grid); thinner bricks / more ranks
    // can exceed that: the host then leaves the fold off for this partition (MdxDecomp::rows_fit)
    if (__popc(mask) > 7) atomicOr(err, 8u);
    cls[g] = (uint8_t)c; owner[g] = (uint8_t)own; shift_code[g] = (uint8_t)code; send_mask[g] = mask;
}

// Compaction of the partition's lists (segment 2q = "send to q", 2q + 1 = "received from q", 2W = local, 2W + 1 = owned), two
// levels (round 4; it was one flag word per atom and segment - 18 x 1 M flags at 8 ranks - and one scan over all of them: 126 us
// of a 320 us repartition).  Membership is a function of what dd_classify_kernel already stored per atom; a wave counts its members
// of every segment with one ballot each, the counts [segment][wave] are scanned (a few hundred thousand words), and the fill
// pass ranks an atom inside its wave with the same ballots: lists in ascending atom order, as before.
__device__ __forceinline__ bool dd_member(int seg, int world, uint32_t c, uint32_t own, uint32_t mask) {
    if (seg < 2 * world) { const int q = seg >> 1; return (seg & 1) ? (c >= 2u && own == (uint32_t)q) : (((mask >> q) & 1u) != 0u); }
    return seg == 2 * world ? c != 0u : c == 1u;
}
__global__ __launch_bounds__(256) void dd_count_kernel(uint32_t N, int world, const uint8_t* __restrict__ cls, const uint8_t* __restrict__ owner,
                                                       const uint32_t* __restrict__ send_mask, uint32_t n_waves, uint32_t* __restrict__ wave_cnt) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = g < N;
    const uint32_t c = in ? cls[g] : 0u, own = in ? owner[g] : 0xFFu, mask = in ? send_mask[g] : 0u;
    const uint32_t wave = g >> 6;
    if (wave >= n_waves) return;
    const int nseg = 2 * world + 2;
    for (int seg = 0; seg < nseg; ++seg) {
        const unsigned long long b = __ballot(in && dd_member(seg, world, c, own, mask));
        if ((threadIdx.x & 63) == 0) wave_cnt[(size_t)seg * n_waves + wave] = (uint32_t)__popcll(b);
    }
    if (g == 0) wave_cnt[(size_t)nseg * n_waves] = 0u;      // the scan's trailing element: the grand total lands there
}

struct DdFill {
    uint32_t seg_start[2 * DD_MAX_WORLD + 3];   // scan value at the head of every segment (+ the grand total)
    uint32_t send_base[DD_MAX_WORLD], recv_base[DD_MAX_WORLD];   // first row of a peer's segment in the halo buffers
};

__global__ __launch_bounds__(256) void dd_fill_kernel(uint32_t N, int world, DdPart p, DdFill f, const uint32_t* __restrict__ send_mask,
                                                      const uint8_t* __restrict__ owner, uint32_t n_waves,
                                                      const uint32_t* __restrict__ scan, const float4* __restrict__ g_pos,
                                                      const float4* __restrict__ g_vel, const uint8_t* __restrict__ cls,
                                                      const uint8_t* __restrict__ shift_code, uint32_t* __restrict__ send_ids,
                                                      uint32_t* __restrict__ recv_ids, float4* __restrict__ recv_shift,
                                                      uint32_t* __restrict__ gid_local, uint8_t* __restrict__ ghost_local,
                                                      float4* __restrict__ pos_l, float4* __restrict__ vel_l, uint32_t* __restrict__ owned_gid,
                                                      float4* __restrict__ pos_at_part) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && (int)threadIdx.x < world && (int)threadIdx.x != p.rank) {
        // the flag row behind every peer's segment (it was three fills over the whole lists and a kernel's worth of gaps)
        const int q = threadIdx.x;
        const uint32_t ns = f.seg_start[2 * q + 1] - f.seg_start[2 * q], nr = f.seg_start[2 * q + 2] - f.seg_start[2 * q + 1];
        send_ids[f.send_base[q] + ns] = MDX_INVALID;
        recv_ids[f.recv_base[q] + nr] = MDX_INVALID; recv_shift[f.recv_base[q] + nr] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool in = g < N;
    const uint32_t c = in ? cls[g] : 0u, own = in ? owner[g] : 0xFFu, mask = in ? send_mask[g] : 0u;
    const uint32_t wave = g >> 6;
    if (wave >= n_waves) return;
    const int lane = threadIdx.x & 63;
    const uint32_t code = in ? shift_code[g] : 0u;
    const float sx = (float)((int)(code & 3u) - 1) * p.len[0], sy = (float)((int)((code >> 2) & 3u) - 1) * p.len[1],
                sz = (float)((int)((code >> 4) & 3u) - 1) * p.len[2];
    const int nseg = 2 * world + 2;
    for (int seg = 0; seg < nseg; ++seg) {
        const bool m = in && dd_member(seg, world, c, own, mask);
        const unsigned long long b = __ballot(m);
        if (!m) continue;
        const uint32_t k = scan[(size_t)seg * n_waves + wave] - f.seg_start[seg] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (seg < 2 * world) {
            const int q = seg >> 1;
            if (!(seg & 1)) send_ids[f.send_base[q] + k] = g;
            else { recv_ids[f.recv_base[q] + k] = g; recv_shift[f.recv_base[q] + k] = make_float4(sx, sy, sz, 0.f); }
        } else if (seg == 2 * world) {
            const float4 pg = g_pos[g];
            gid_local[k] = g; ghost_local[k] = c == 2u ? 1 : (c == 3u ? 3 : 0);   // bit 0: ghost, bit 1: no pair interactions here
            const float4 pl = make_float4(dd_wrap1(pg.x, p.lo[0], p.len[0]) + sx, dd_wrap1(pg.y, p.lo[1], p.len[1]) + sy,
                                          dd_wrap1(pg.z, p.lo[2], p.len[2]) + sz, 0.f);
            pos_l[k] = pl; pos_at_part[k] = pl;
            vel_l[k] = g_vel[g];
        } else owned_gid[k] = g;
    }
}

__global__ void dd_seg_heads_kernel(int nseg, uint32_t n_waves, const uint32_t* __restrict__ scan, uint32_t* __restrict__ out) {
    const int s = threadIdx.x;
    if (s <= nseg) out[s] = scan[(size_t)s * n_waves];
}

// owned rows of the global gather: (x, y, z, global id), (vx, vy, vz, -)[, (fx, fy, fz, -)]
__global__ __launch_bounds__(256) void dd_gather_pack_kernel(uint32_t n_owned, int rows, const uint32_t* __restrict__ owned_gid,
                                                             const uint32_t* __restrict__ slot_of, const float4* __restrict__ posq,
                                                             const float4* __restrict__ vel, const float4* __restrict__ force,
                                                             float4* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_owned) return;
    const uint32_t g = owned_gid[k], s = slot_of[g];
    float4 p = posq[s]; p.w = __uint_as_float(g);
    out[(size_t)rows * k] = p;
    out[(size_t)rows * k + 1] = vel[s];
    if (rows == 3) out[(size_t)rows * k + 2] = force[s];
}
__global__ __launch_bounds__(256) void dd_gather_scatter_kernel(uint32_t n, int rows, const float4* __restrict__ in, float4* __restrict__ g_pos,
                                                                float4* __restrict__ g_vel, float4* __restrict__ g_frc, uint32_t N,
                                                                uint32_t* __restrict__ err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float4 p = in[(size_t)rows * k];
    const uint32_t g = __float_as_uint(p.w);
    if (g >= N) { atomicOr(err, 2u); return; }
    p.w = 0.f;
    g_pos[g] = p;
    float4 v = in[(size_t)rows * k + 1]; v.w = 0.f;
    g_vel[g] = v;
    if (rows == 3) g_frc[g] = in[(size_t)rows * k + 2];
}

// largest squared displacement of a local atom since the last repartition (uncut dimensions stay periodic inside the engine).
// Grid-stride over at most 64 workgroups, one atomic per workgroup: one per wave - 3.2 k contended atomics for one rank of 8 of
// the 1 M-atom box - made this a 32 us kernel (it streams 8 MB)
__global__ __launch_bounds__(256) void dd_drift_kernel(uint32_t n_local, const uint32_t* __restrict__ gid_local, const uint32_t* __restrict__ slot_of,
                                                       const float4* __restrict__ posq, const float4* __restrict__ pos_at_part, DdPart p,
                                                       uint32_t* __restrict__ out_bits) {
    __shared__ float s_max[4];
    float d2m = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_local; i += gridDim.x * blockDim.x) {
        const uint32_t s = slot_of[gid_local[i]];
        if (s == MDX_INVALID) continue;
        const float4 a = posq[s], b = pos_at_part[i];
        float d[3] = {a.x - b.x, a.y - b.y, a.z - b.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (p.grid[k] == 1) d[k] -= rintf(d[k] / p.len[k]) * p.len[k];
        float d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (!(d2 < 1.0e30f)) d2 = 3.0e38f;
        d2m = fmaxf(d2m, d2);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) d2m = fmaxf(d2m, __shfl_xor(d2m, m));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = d2m;
    __syncthreads();
    if (threadIdx.x == 0) {
        d2m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (d2m > 0.f) atomicMax(out_bits, __float_as_uint(d2m));
    }
}

// ------------------------------------------------------------------------------------------------------------------
static void process_grid(int world, int g[3]) {
    if (world == 1) { g[0] = g[1] = g[2] = 1; return; }
    if (world == 2) { g[0] = 2; g[1] = 1; g[2] = 1; return; }
    if (world == 4) { g[0] = 2; g[1] = 2; g[2] = 1; return; }
    if (world == 8) { g[0] = 2; g[1] = 2; g[2] = 2; return; }
    int v[3] = {1, 1, 1};
    int n = world;
    for (int f = 2; n > 1; ++f)
        while (n % f == 0) { int m = 0; for (int d = 1; d < 3; ++d) if (v[d] < v[m]) m = d; v[m] *= f; n /= f; }
    std::sort(v, v + 3, [](int a, int b) { return a > b; });
    g[0] = v[0]; g[1] = v[1]; g[2] = v[2];
}

template <typename T>
static int dd_alloc(T** p, size_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIP_TRY(hipMalloc((void**)p, std::max<size_t>(sizeof(T) * n, 16)));
    return MDX_OK;
}

static DdPart make_part(const MdxDecomp* dd) {
    DdPart p{};
    for (int d = 0; d < 3; ++d) { p.lo[d] = dd->box_lo[d]; p.len[d] = dd->box_len[d]; p.grid[d] = dd->grid[d]; p.coord[d] = dd->coord[d]; }
    p.rank = dd->rank; p.world = dd->world; p.halo = dd->halo;
    return p;
}

// Owners, ghosts, image shifts and both halo lists from g_pos / g_vel; the engine is told its new local atom set.
static int dd_partition(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    const uint32_t N = h->N; const int W = dd->world;
    hipStream_t st = h->stream;
    const DdPart p = make_part(dd);
    const int nseg = 2 * W + 2;
    const uint32_t n_waves = div_up(N, 64);
    const size_t nflags = (size_t)nseg * n_waves + 1;      // per-wave member counts of every segment (+ the scan's trailing element)
    if (nflags > dd->cap_flags) {
        MDX_TRY(dd_alloc(&dd->flags, nflags)); MDX_TRY(dd_alloc(&dd->scan, nflags));
        MDX_TRY(dd_alloc(&dd->scan_sums, nflags / 2048 + 64));
        dd->cap_flags = nflags;
    }
    HIP_TRY(hipMemsetAsync(h->d.flags_dev, 0, sizeof(uint32_t) * 4, st));
    hipLaunchKernelGGL(dd_classify_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, N, dd->g_pos, dd->anchor, p, dd->half_shell ? 1 : 0,
                       h->n_roles ? h->d.role_off_o : nullptr, h->d.role_rec_o, dd->cls, dd->owner,
                       dd->shift_code, dd->send_mask, h->d.flags_dev);
    hipLaunchKernelGGL(dd_count_kernel, dim3(div_up(n_waves * 64u, 256)), dim3(256), 0, st, N, W, dd->cls, dd->owner, dd->send_mask, n_waves, dd->flags);
    MDX_TRY(mdx_exclusive_scan_u32_ex(h, dd->flags, dd->scan, (uint32_t)nflags, dd->scan_sums));
    uint32_t* d_heads = (uint32_t*)dd->red;     // 64 doubles = 128 words of scratch
    hipLaunchKernelGGL(dd_seg_heads_kernel, dim3(1), dim3(128), 0, st, nseg, n_waves, dd->scan, d_heads);
    uint32_t heads[2 * DD_MAX_WORLD + 3], err[4];
    HIP_TRY(hipMemcpyAsync(heads, d_heads, sizeof(uint32_t) * (nseg + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(err, h->d.flags_dev, sizeof(err), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (err[0] & 1u) FAIL(MDX_EPARAM, "decomposition: an owned atom lies outside its rank's halo region (constraint cluster larger than the halo allows)");
    if (err[0] & 4u) {
        float4 pa{}, pb{};
        (void)hipMemcpy(&pa, dd->g_pos + err[1], sizeof(float4), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&pb, dd->g_pos + err[2], sizeof(float4), hipMemcpyDeviceToHost);
        char buf[320];
        snprintf(buf, sizeof(buf), "decomposition: a bonded partner of an owned atom lies beyond the halo (atom %u at %.2f %.2f %.2f, partner %u at %.2f %.2f %.2f, "
                 "%s rank %u, this rank %d, halo %.2f)", err[1], pa.x, pa.y, pa.z, err[2], pb.x, pb.y, pb.z, (err[3] & 0x100u) ? "partner owned by" : "atom owned by",
                 err[3] & 0xFFu, dd->rank, dd->halo);
        FAIL(MDX_EPARAM, buf);
    }
    dd->rows_fit = !(err[0] & 8u);      // (a rank-local matter: the messages are the same whoever packs them)
    DdFill f{};
    for (int s = 0; s <= nseg; ++s) f.seg_start[s] = heads[s];
    dd->send_segs.clear(); dd->recv_segs.clear();
    uint32_t s0 = 0, r0 = 0;
    for (int q = 0; q < W; ++q) {
        if (q == dd->rank) continue;
        const uint32_t ns = heads[2 * q + 1] - heads[2 * q], nr = heads[2 * q + 2] - heads[2 * q + 1];
        f.send_base[q] = s0; f.recv_base[q] = r0;
        dd->send_segs.push_back({q, s0, ns + 1}); dd->recv_segs.push_back({q, r0, nr + 1});   // + the flag row
        s0 += ns + 1; r0 += nr + 1;
    }
    dd->n_send = s0; dd->n_recv = r0;
    dd->n_local = heads[2 * W + 1] - heads[2 * W];
    dd->n_owned = heads[2 * W + 2] - heads[2 * W + 1];
    if (dd->n_local == 0) FAIL(MDX_EPARAM, "decomposition: a rank received no atoms (empty brick)");
    if (s0 > dd->cap_send) {
        dd->cap_send = s0 + s0 / 4 + 64;
        MDX_TRY(dd_alloc(&dd->send_ids, dd->cap_send)); MDX_TRY(dd_alloc(&dd->send_buf, dd->cap_send)); MDX_TRY(dd_alloc(&dd->frc_recv, dd->cap_send));
        // (a transport that delivers writes every row before it is read; the null transport with a stated wire time reads the rows as
        // they are - the force call in front of the first halo exchange added whatever the allocation held to the owned atoms' forces)
        HIP_TRY(hipMemsetAsync(dd->frc_recv, 0, sizeof(float4) * (size_t)dd->cap_send, st));
    }
    if (r0 > dd->cap_recv) {
        dd->cap_recv = r0 + r0 / 4 + 64;
        MDX_TRY(dd_alloc(&dd->recv_ids, dd->cap_recv)); MDX_TRY(dd_alloc(&dd->recv_buf, dd->cap_recv)); MDX_TRY(dd_alloc(&dd->recv_shift, dd->cap_recv));
        MDX_TRY(dd_alloc(&dd->frc_send, dd->cap_recv));
        HIP_TRY(hipMemsetAsync(dd->recv_buf, 0, sizeof(float4) * (size_t)dd->cap_recv, st));
    }
    // the engine's own arrays are the partition's output (gid, ghost flags, positions, velocities of the local atoms in list order)
    dd->gid_local = h->d.gid; dd->ghost_local = h->d.lflag; dd->pos_l = h->d.pos_orig; dd->vel_l = h->d.vel_orig;
    hipLaunchKernelGGL(dd_fill_kernel, dim3(div_up(n_waves * 64u, 256)), dim3(256), 0, st, N, W, p, f, dd->send_mask, dd->owner, n_waves, dd->scan, dd->g_pos, dd->g_vel,
                       dd->cls, dd->shift_code, dd->send_ids, dd->recv_ids, dd->recv_shift, dd->gid_local, dd->ghost_local, dd->pos_l,
                       dd->vel_l, dd->owned_gid, dd->pos_at_part);
    HIP_TRY(hipGetLastError());
    // the local region: brick + halo (+ a little room) in cut dimensions, the whole box in the others
    float lo[3], hi[3];
    int per_mask = 0x10;
    for (int d = 0; d < 3; ++d) {
        if (dd->grid[d] > 1) { lo[d] = dd->brick_lo[d] - dd->halo - 1.0f; hi[d] = dd->brick_hi[d] + dd->halo + 1.0f; }
        else { lo[d] = dd->box_lo[d]; hi[d] = dd->box_lo[d] + dd->box_len[d]; per_mask |= 1 << d; }
    }
    MDX_TRY(mdx_set_local_atoms_impl(h, dd->n_local, dd->gid_local, dd->ghost_local, (const float*)dd->pos_l, (const float*)dd->vel_l,
                                     lo, hi, per_mask));
    dd->repartitions++;
    dd->local_rebuilds_since = 0;
    dd->rows_valid = false;
    return MDX_OK;
}

// Every RCCL operation of a handle is issued on ONE stream - its communication stream - so that the communicator sees one
// ordered sequence of operations whatever the compute streams are doing.  These two bracket an operation that consumes
// data produced on the compute stream and whose result the compute stream consumes.
static int dd_comm_enter(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    if (dd->comm_stream == h->stream) return MDX_OK;
    HIP_TRY(hipEventRecord(dd->ev_packed, h->stream));
    HIP_TRY(hipStreamWaitEvent(dd->comm_stream, dd->ev_packed, 0));
    return MDX_OK;
}
static int dd_comm_leave(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    if (dd->comm_stream == h->stream) return MDX_OK;
    HIP_TRY(hipEventRecord(dd->ev_arrived, dd->comm_stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, dd->ev_arrived, 0));
    return MDX_OK;
}
int mdx_dd_allreduce_dev(mdx_handle* h, double* dev, size_t n) {
    MdxDecomp* dd = h->dd;
    if (!dd || dd->world == 1 || n == 0) return MDX_OK;
    MDX_TRY(dd_comm_enter(h));
    for (size_t k = 0; k < n; k += 4096) MDX_TRY(dd->tr->all_reduce(dev + k, std::min<size_t>(4096, n - k), 0, dd->comm_stream));
    MDX_TRY(dd_comm_leave(h));
    return MDX_OK;
}
int mdx_dd_allreduce_f32(mdx_handle* h, float* dev, size_t n, hipStream_t produced_on) {
    MdxDecomp* dd = h->dd;
    if (dd->comm_stream == produced_on) return dd->tr->all_reduce_f32(dev, n, produced_on);
    HIP_TRY(hipEventRecord(dd->ev_packed, produced_on));
    HIP_TRY(hipStreamWaitEvent(dd->comm_stream, dd->ev_packed, 0));
    MDX_TRY(dd->tr->all_reduce_f32(dev, n, dd->comm_stream));
    HIP_TRY(hipEventRecord(dd->ev_arrived, dd->comm_stream));
    HIP_TRY(hipStreamWaitEvent(produced_on, dd->ev_arrived, 0));
    return MDX_OK;
}

// A send/recv group on the handle's communication stream, consuming data produced on `produced_on` and producing data that
// stream consumes (the slab-decomposed SPME chain: mesh redistribution and FFT transposes).
int mdx_dd_exchange(mdx_handle* h, const float4* send, const std::vector<MdxSeg>& ssegs, float4* recv, const std::vector<MdxSeg>& rsegs,
                    hipStream_t produced_on) {
    MdxDecomp* dd = h->dd;
    if (dd->comm_stream == produced_on) return dd->tr->exchange(send, ssegs, recv, rsegs, produced_on);
    HIP_TRY(hipEventRecord(dd->ev_packed, produced_on));
    HIP_TRY(hipStreamWaitEvent(dd->comm_stream, dd->ev_packed, 0));
    MDX_TRY(dd->tr->exchange(send, ssegs, recv, rsegs, dd->comm_stream));
    HIP_TRY(hipEventRecord(dd->ev_arrived, dd->comm_stream));
    HIP_TRY(hipStreamWaitEvent(produced_on, dd->ev_arrived, 0));
    return MDX_OK;
}

int mdx_dd_gather_global(mdx_handle* h, bool with_force) {
    MdxDecomp* dd = h->dd;
    hipStream_t st = h->stream;
    const int W = dd->world, R = with_force ? 3 : 2;
    const uint32_t N = h->N;
    if (!h->in_slot_space) MDX_TRY(mdx_rebuild(h));
    if ((size_t)R * dd->n_owned > dd->cap_gat_send) { dd->cap_gat_send = (size_t)3 * dd->n_owned + 1024; MDX_TRY(dd_alloc(&dd->gat_send, dd->cap_gat_send)); }
    if ((size_t)R * N > dd->cap_gat_recv) { dd->cap_gat_recv = (size_t)3 * N; MDX_TRY(dd_alloc(&dd->gat_recv, dd->cap_gat_recv)); }
    if (with_force && !dd->g_frc) MDX_TRY(dd_alloc(&dd->g_frc, N));
    hipLaunchKernelGGL(dd_gather_pack_kernel, dim3(div_up(std::max(dd->n_owned, 1u), 256)), dim3(256), 0, st, dd->n_owned, R, dd->owned_gid,
                       h->d.slot_of, h->d.posq, h->d.vel, h->d.force, dd->gat_send);
    uint32_t counts[DD_MAX_WORLD];
    MDX_TRY(dd_comm_enter(h));
    MDX_TRY(dd->tr->all_gather_u32(dd->n_owned, counts, dd->comm_stream));
    if (!dd->tr->delivers()) {   // one rank of N alone: the others' rows keep their last known values
        MDX_TRY(dd_comm_leave(h));
        hipLaunchKernelGGL(dd_gather_scatter_kernel, dim3(div_up(std::max(dd->n_owned, 1u), 256)), dim3(256), 0, st, dd->n_owned, R, dd->gat_send,
                           dd->g_pos, dd->g_vel, dd->g_frc, N, h->d.flags_dev);
        HIP_TRY(hipGetLastError());
        return MDX_OK;
    }
    uint64_t tot = 0;
    std::vector<MdxSeg> ss, rs;
    uint32_t my_off = 0;
    for (int q = 0; q < W; ++q) {
        if (q == dd->rank) my_off = (uint32_t)tot;
        else { ss.push_back({q, 0u, (uint32_t)R * dd->n_owned}); rs.push_back({q, (uint32_t)(R * tot), (uint32_t)R * counts[q]}); }
        tot += counts[q];
    }
    if (tot != N) FAIL(MDX_EDEVICE, "decomposition: the ranks' owned atoms do not add up to the system (an atom was lost or duplicated)");
    HIP_TRY(hipMemcpyAsync(dd->gat_recv + (size_t)R * my_off, dd->gat_send, sizeof(float4) * R * dd->n_owned, hipMemcpyDeviceToDevice, dd->comm_stream));
    MDX_TRY(dd->tr->exchange(dd->gat_send, ss, dd->gat_recv, rs, dd->comm_stream));
    MDX_TRY(dd_comm_leave(h));
    HIP_TRY(hipMemsetAsync(h->d.flags_dev, 0, sizeof(uint32_t) * 4, st));
    hipLaunchKernelGGL(dd_gather_scatter_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, N, R, dd->gat_recv, dd->g_pos, dd->g_vel, dd->g_frc, N,
                       h->d.flags_dev);
    HIP_TRY(hipGetLastError());
    uint32_t err[4];
    HIP_TRY(hipMemcpyAsync(err, h->d.flags_dev, sizeof(err), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (err[0] & 2u) FAIL(MDX_EDEVICE, "decomposition: a gathered row carries an atom id outside the system (corrupted message)");
    return MDX_OK;
}

int mdx_dd_allreduce_host(mdx_handle* h, double* v, int n, bool max_u32) {
    MdxDecomp* dd = h->dd;
    if (!dd || dd->world == 1 || n <= 0) return MDX_OK;
    if (n > 64) FAIL(MDX_EPARAM, "internal: small all-reduce only");
    HIP_TRY(hipStreamSynchronize(h->stream));      // (the values came from a synchronised read-back; the scratch is free)
    hipStream_t st = dd->comm_stream;
    if (max_u32) {
        uint32_t w[64];
        for (int k = 0; k < n; ++k) w[k] = (uint32_t)v[k];
        HIP_TRY(hipMemcpyAsync(dd->red, w, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
        MDX_TRY(dd->tr->all_reduce(dd->red, n, 1, st));
        HIP_TRY(hipMemcpyAsync(w, dd->red, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int k = 0; k < n; ++k) v[k] = (double)w[k];
    } else {
        HIP_TRY(hipMemcpyAsync(dd->red, v, sizeof(double) * n, hipMemcpyHostToDevice, st));
        MDX_TRY(dd->tr->all_reduce(dd->red, n, 0, st));
        HIP_TRY(hipMemcpyAsync(v, dd->red, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return MDX_OK;
}

// ---- per-step halo ------------------------------------------------------------------------------------------------
static int dd_pipe_tables(mdx_handle* h);
int mdx_dd_halo_begin(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    if (dd->world == 1) return MDX_OK;
    const int fw = dd->halo_step >= 0 ? dd->halo_step + 1 : -1;
    if (dd->tr->loopback() && !dd->rows_valid && h->in_slot_space) MDX_TRY(dd_pipe_tables(h));      // (keeps the loop-back buffers current)
    if (dd->packed_by_drift) dd->packed_by_drift = false;      // the fused bonded + kick + drift pass has filled the message (flag rows included)
    else {
        mdx_prof_begin(h, 6);
        const int rc_pack = mdx_pack_positions(h, dd->send_ids, dd->n_send, (float*)dd->send_buf, fw);
        mdx_prof_end(h);
        MDX_TRY(rc_pack);
    }
    if (dd->comm_stream != h->stream) {
        HIP_TRY(hipEventRecord(dd->ev_packed, h->stream));
        HIP_TRY(hipStreamWaitEvent(dd->comm_stream, dd->ev_packed, 0));
    }
    mdx_prof_begin(h, 7, dd->comm_stream);
    const int rc_x = dd->tr->exchange(dd->send_buf, dd->send_segs, dd->recv_buf, dd->recv_segs, dd->comm_stream);
    mdx_prof_end(h);
    MDX_TRY(rc_x);
    if (dd->comm_stream != h->stream) HIP_TRY(hipEventRecord(dd->ev_arrived, dd->comm_stream));
    return MDX_OK;
}

int mdx_dd_halo_end(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    if (dd->world == 1) return MDX_OK;
    const int fw = dd->halo_step >= 0 ? dd->halo_step + 1 : -1;
    if (dd->comm_stream != h->stream) HIP_TRY(hipStreamWaitEvent(h->stream, dd->ev_arrived, 0));
    if (dd->tr->delivers() || dd->tr->loopback()) {
        mdx_prof_begin(h, 8);
        const int rc_u = mdx_unpack_positions(h, dd->recv_ids, dd->n_recv, (const float*)dd->recv_buf, (const float*)dd->recv_shift, fw);
        mdx_prof_end(h);
        MDX_TRY(rc_u);
    }
    return MDX_OK;
}

// ---- half shell: the forces a rank computed on its ghosts go back to their owners -------------------------------------
// Every cross-rank pair is evaluated once, on the rank that keeps the other atom as a ghost; the half-list pair kernel leaves
// the reaction in the ghost's row of the force array.  Rows travel along the halo segments in the opposite direction (what
// was received from q is sent to q), flag rows included: a peer that found its list stale during this step's drift tells the
// ranks BELOW it here (the position message only reaches the ranks above), so every later kernel of the chunk is gated off
// everywhere.  The owner adds the rows with f32 atomics: an atom may be a ghost on several peers.
__global__ void dd_pack_force_kernel(uint32_t n, const uint32_t* __restrict__ atom_idx, const uint32_t* __restrict__ slot_of,
                                     const float4* __restrict__ force, float4* __restrict__ out, const uint32_t* __restrict__ flag_word) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = atom_idx[i];
    if (g == MDX_INVALID) { out[i] = make_float4(flag_word ? __uint_as_float(*flag_word) : 0.f, 0.f, 0.f, 0.f); return; }
    const uint32_t s = slot_of[g];
    out[i] = (s == MDX_INVALID) ? make_float4(0.f, 0.f, 0.f, 0.f) : force[s];
}
__global__ void dd_add_force_kernel(uint32_t n, const uint32_t* __restrict__ atom_idx, const uint32_t* __restrict__ slot_of,
                                    float4* __restrict__ force, const float4* __restrict__ in, uint32_t* __restrict__ flag_word) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = atom_idx[i];
    if (g == MDX_INVALID) { if (flag_word) atomicMax(flag_word, __float_as_uint(in[i].x)); return; }
    const uint32_t s = slot_of[g];
    if (s == MDX_INVALID) return;
    const float4 f = in[i];
    float* const dst = reinterpret_cast<float*>(force + s);
    unsafeAtomicAdd(dst, f.x); unsafeAtomicAdd(dst + 1, f.y); unsafeAtomicAdd(dst + 2, f.z);
}

bool mdx_dd_half_shell(const mdx_handle* h) { return h->dd && h->dd->half_shell && h->dd->world > 1; }
// the brick of this rank in a cut dimension (false: the dimension is not cut, or the handle is not decomposed)
bool mdx_dd_brick_bounds(const mdx_handle* h, int d, float* lo, float* hi) {
    if (!h->dd || h->dd->grid[d] <= 1) return false;
    *lo = h->dd->brick_lo[d]; *hi = h->dd->brick_hi[d];
    return true;
}

int mdx_dd_force_return_begin(mdx_handle* h, int flag_word) {
    MdxDecomp* dd = h->dd;
    if (!mdx_dd_half_shell(h) || !h->in_slot_space) return MDX_OK;
    hipStream_t st = h->stream;
    const uint32_t* fw = flag_word >= 0 ? &h->d.ctl->disp2[flag_word] : nullptr;
    mdx_prof_begin(h, 9);
    if (dd->n_recv)
        hipLaunchKernelGGL(dd_pack_force_kernel, dim3(div_up(dd->n_recv, 256)), dim3(256), 0, st, dd->n_recv, dd->recv_ids, h->d.slot_of,
                           h->d.force, dd->frc_send, fw);
    mdx_prof_end(h);
    HIP_TRY(hipGetLastError());
    if (dd->comm_stream != st) {
        HIP_TRY(hipEventRecord(dd->ev_packed, st));
        HIP_TRY(hipStreamWaitEvent(dd->comm_stream, dd->ev_packed, 0));
    }
    mdx_prof_begin(h, 10, dd->comm_stream);
    const int rc_x = dd->tr->exchange(dd->frc_send, dd->recv_segs, dd->frc_recv, dd->send_segs, dd->comm_stream);
    mdx_prof_end(h);
    MDX_TRY(rc_x);
    if (dd->comm_stream != st) HIP_TRY(hipEventRecord(dd->ev_arrived, dd->comm_stream));
    dd->force_return_pending = true;
    return MDX_OK;
}

int mdx_dd_force_return_end(mdx_handle* h, int flag_word) {
    MdxDecomp* dd = h->dd;
    if (!dd || !dd->force_return_pending) return MDX_OK;
    dd->force_return_pending = false;
    hipStream_t st = h->stream;
    if (dd->comm_stream != st) HIP_TRY(hipStreamWaitEvent(st, dd->ev_arrived, 0));
    // the next step's fused bonded + kick + drift pass adds the rows (and merges the peers' stale words) itself
    if (dd->pipe_now && h->bonded_deferred && flag_word >= 0) { dd->frc_deferred = true; return MDX_OK; }
    uint32_t* fw = flag_word >= 0 ? &h->d.ctl->disp2[flag_word] : nullptr;
    if ((dd->tr->delivers() || dd->tr->loopback()) && dd->n_send) {
        mdx_prof_begin(h, 11);
        hipLaunchKernelGGL(dd_add_force_kernel, dim3(div_up(dd->n_send, 256)), dim3(256), 0, st, dd->n_send, dd->send_ids, h->d.slot_of,
                           h->d.force, dd->frc_recv, fw);
        mdx_prof_end(h);
    }
    HIP_TRY(hipGetLastError());
    return MDX_OK;
}

// ---- what the fused bonded + kick + drift pass of a decomposed step carries ("fold") ----------------------------------------------------
// The plain arrangement of a step is a chain on one stream: drift -> pack -> [send/recv] -> unpack -> pair kernel -> ghost-force pack ->
// [send/recv] -> add, with ~4 us between dependent launches.  In the steps of the fused bonded + kick + drift pass (mdx_integrate.hip)
// that pass takes the pack and the add with it: a slot knows its rows of the position message (at most seven, one per peer), writes its
// new position into them, and adds the rows of the force message - same layout - that came back for it after the previous force call;
// the workgroup that finishes last writes the flag rows.  Two kernels and two launch boundaries less per step.
// (Round 5 also built and measured the arrangement the round-4 verdict sketched - ONE pair launch on a side stream whose boundary
// workgroups wait for a word the unpack kernel publishes, both messages on the compute stream - and dropped it: DESIGN_HISTORY.md
// "Round 5", profiles/r05_pipe_experiments.txt.)
__global__ void pipe_rows_kernel(uint32_t n_send, const uint32_t* __restrict__ send_ids, const uint32_t* __restrict__ slot_of,
                                 uint32_t* __restrict__ send_cnt, uint32_t* __restrict__ send_rows, uint32_t* err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_send) return;
    const uint32_t g = send_ids[i];
    if (g == MDX_INVALID) return;            // a flag row
    const uint32_t s = slot_of[g];
    if (s == MDX_INVALID) return;
    const uint32_t k = atomicAdd(send_cnt + s, 1u);
    if (k < 7u) send_rows[(size_t)s * 7 + k] = i; else atomicOr(err, 1u);      // (cannot happen while MdxDecomp::rows_fit gates the fold; a word of its own in PipeCtl)
}
// null transport with a stated wire time: the receive buffers hold what the unpack / add kernels turn into no-ops
__global__ void pipe_loopback_kernel(uint32_t n_recv, const uint32_t* __restrict__ recv_ids, const uint32_t* __restrict__ slot_of,
                                     const float4* __restrict__ posq, const float4* __restrict__ shift, float4* __restrict__ recv_buf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_recv) return;
    const uint32_t g = recv_ids[i];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g != MDX_INVALID) {
        const uint32_t s = slot_of[g];
        if (s != MDX_INVALID) { v = posq[s]; const float4 sh = shift[i]; v.x -= sh.x; v.y -= sh.y; v.z -= sh.z; }
    }
    recv_buf[i] = v;
}

void mdx_dd_note_rebuild(mdx_handle* h) { if (h->dd) h->dd->rows_valid = false; }
void mdx_dd_pipe_chunk_end(mdx_handle* h) { if (h->dd) { h->dd->pipe_now = false; h->dd->frc_deferred = false; h->dd->packed_by_drift = false; } }

// per-slot rows of the position message, tile counts of the pair launch; (null transport with a wire time) the loop-back buffers
static int dd_pipe_tables(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    hipStream_t st = h->stream;
    if (!dd->pipe_ctl) {
        HIP_TRY(hipMalloc((void**)&dd->pipe_ctl, sizeof(PipeCtl)));
        HIP_TRY(hipMemsetAsync(dd->pipe_ctl, 0, sizeof(PipeCtl), st));
    }
    if (dd->cap_rows_slots < h->S) {
        dd->cap_rows_slots = h->cap_tiles * MDX_TILE;
        MDX_TRY(dd_alloc(&dd->send_cnt, dd->cap_rows_slots)); MDX_TRY(dd_alloc(&dd->send_rows, (size_t)dd->cap_rows_slots * 7));
    }
    HIP_TRY(hipMemsetAsync(dd->send_cnt, 0, sizeof(uint32_t) * (size_t)h->S, st));
    if (dd->n_send)
        hipLaunchKernelGGL(pipe_rows_kernel, dim3(div_up(dd->n_send, 256)), dim3(256), 0, st, dd->n_send, dd->send_ids, h->d.slot_of, dd->send_cnt,
                           dd->send_rows, &dd->pipe_ctl->rows_overflow);
    if (dd->tr->loopback()) {
        if (dd->n_recv)
            hipLaunchKernelGGL(pipe_loopback_kernel, dim3(div_up(dd->n_recv, 256)), dim3(256), 0, st, dd->n_recv, dd->recv_ids, h->d.slot_of, h->d.posq,
                               dd->recv_shift, dd->recv_buf);
        if (dd->n_send) HIP_TRY(hipMemsetAsync(dd->frc_recv, 0, sizeof(float4) * (size_t)dd->n_send, st));
    }
    HIP_TRY(hipGetLastError());
    dd->rows_valid = true;
    return MDX_OK;
}

// the fused bonded + kick + drift pass about to be launched: returned ghost forces in (if the last force call left them to it), halo
// pack out (if this step's drift carries it)
bool mdx_dd_pipe_fill(mdx_handle* h, FusedArgs& a, uint32_t* gate_word) {
    MdxDecomp* dd = h->dd;
    a.pipe_flags = 0u;
    if (!dd || !(dd->pipe_now || dd->frc_deferred)) return false;
    if (!dd->rows_valid && dd_pipe_tables(h) != MDX_OK) {
        // no tables, no fold: the plain pass runs instead.  Ghost forces that mdx_dd_force_return_end left to this pass are added by
        // the kernel of the unfolded arrangement first (gate_word: the control word of that force call, where the peers' stale words
        // merge), and the step's halo goes out through the pack kernel (pipe_now off: mdx_dd_halo_begin packs)
        if (dd->frc_deferred && (dd->tr->delivers() || dd->tr->loopback()) && dd->n_send)
            hipLaunchKernelGGL(dd_add_force_kernel, dim3(div_up(dd->n_send, 256)), dim3(256), 0, h->stream, dd->n_send, dd->send_ids, h->d.slot_of,
                               h->d.force, dd->frc_recv, gate_word);
        dd->frc_deferred = false; dd->pipe_now = false; dd->fold_ok = false;
        return false;
    }
    a.pipe_flags = (dd->frc_deferred ? 1u : 0u) | (dd->pipe_now ? 2u : 0u);
    dd->frc_deferred = false;
    a.send_cnt = dd->send_cnt; a.send_rows = dd->send_rows; a.frc_in = dd->frc_recv; a.send_buf = dd->send_buf;
    a.n_flag = 0;
    static_assert(sizeof(a.flag_rows) / sizeof(a.flag_rows[0]) >= DD_MAX_WORLD, "one flag row per peer");
    for (const MdxSeg& sg : dd->send_segs) if (sg.nrows) a.flag_rows[a.n_flag++] = sg.row0 + sg.nrows - 1;      // EVERY peer must hear "my list went stale"
    a.pc = dd->pipe_ctl; a.gate_word = gate_word;
    if (dd->pipe_now) { ++dd->pipe_gen; ++dd->pipe_steps; dd->packed_by_drift = true; }
    return true;
}

// ---- stale list: local rebuild or repartition (the same branch on every rank) --------------------------------------
static int dd_enqueue_drift_probe(mdx_handle* h, uint32_t* bits) {
    MdxDecomp* dd = h->dd;
    hipStream_t st = h->stream;
    HIP_TRY(hipMemsetAsync(bits, 0, sizeof(uint32_t), st));
    if (h->in_slot_space) {
        hipLaunchKernelGGL(dd_drift_kernel, dim3(std::min(div_up(dd->n_local, 256), 64u)), dim3(256), 0, st, dd->n_local, dd->gid_local, h->d.slot_of,
                           h->d.posq, dd->pos_at_part, make_part(dd), bits);
        // Chunk-end probe behind fused bonded + kick + drift passes: every enqueued pass swaps the host's posq / posq_alt pointers, the
        // gated-off ones included, while on the device the passes behind a stale step write nothing - with an odd number of them
        // behind the stale step h->d.posq names the buffer that still holds the positions of the step BEFORE it (mdx_step repoints
        // it after the read-back).  The two buffers hold the last two states: the larger of their drifts is the stale step's or
        // an upper bound within one step of it - never an under-estimate that would approve a local rebuild on an incomplete set.
        if (dd->probe_both_buffers && h->d.posq_alt)
            hipLaunchKernelGGL(dd_drift_kernel, dim3(std::min(div_up(dd->n_local, 256), 64u)), dim3(256), 0, st, dd->n_local, dd->gid_local, h->d.slot_of,
                               h->d.posq_alt, dd->pos_at_part, make_part(dd), bits);
    }
    MDX_TRY(dd_comm_enter(h));
    MDX_TRY(dd->tr->all_reduce(bits, 1, 1, dd->comm_stream));
    MDX_TRY(dd_comm_leave(h));
    return MDX_OK;
}
static bool dd_drift_allows_local_rebuild(const MdxDecomp* dd, uint32_t bits) {
    float d2; std::memcpy(&d2, &bits, 4);
    return dd->margin > 0.f && dd->local_rebuilds_since < 256 && std::sqrt(d2) <= 0.5f * dd->margin - 0.05f;
}

// Step loop, end of every chunk of a decomposed handle: the probe rides behind the chunk's kernels (a ~5 us kernel and a one-word
// all-reduce every ~16 steps) and its word reaches the host with the step-control block.  Skipped with virtual sites: their
// positions at the stale step are only constructed by mdx_dd_on_stale.
int mdx_dd_chunk_end_probe(mdx_handle* h, const uint32_t** word_out) {
    MdxDecomp* dd = h->dd;
    *word_out = nullptr;
    dd->spec_valid = false;
    static const bool on = [] { const char* e = std::getenv("MDX_DD_SPEC_PROBE"); return !(e && e[0] == '0'); }();
    if (!on || dd->world == 1 || h->n_vsites || !h->in_slot_space) return MDX_OK;
    if (!dd->drift_bits) { HIP_TRY(hipMalloc((void**)&dd->drift_bits, 16)); }
    MDX_TRY(dd_enqueue_drift_probe(h, dd->drift_bits));
    *word_out = dd->drift_bits;
    return MDX_OK;
}

static int dd_local_set_still_valid(mdx_handle* h, bool* valid) {
    MdxDecomp* dd = h->dd;
    *valid = true;
    if (dd->world == 1) return MDX_OK;
    if (dd->spec_valid) {      // measured and all-reduced behind the chunk that went stale: the same word on every rank
        dd->spec_valid = false;
        *valid = dd_drift_allows_local_rebuild(dd, dd->spec_bits);
        // MDX_DD_SPEC_CHECK=1 (tests): measure again now - mdx_step has repointed posq at the stale step's buffer - and insist that
        // the word that rode behind the chunk was not smaller (collective: every rank runs the same check)
        const char* chk = std::getenv("MDX_DD_SPEC_CHECK");
        if (chk && chk[0] == '1') {
            uint32_t* bits = (uint32_t*)dd->red;
            MDX_TRY(dd_enqueue_drift_probe(h, bits));
            uint32_t b = 0;
            HIP_TRY(hipMemcpyAsync(&b, bits, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            dd->spec_checks++;
            if (dd->spec_bits < b) FAIL(MDX_EDEVICE, "internal: the chunk-end drift probe under-reported the drift since the last repartition");
        }
        return MDX_OK;
    }
    uint32_t* bits = (uint32_t*)dd->red;
    MDX_TRY(dd_enqueue_drift_probe(h, bits));
    uint32_t b = 0;
    HIP_TRY(hipMemcpyAsync(&b, bits, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *valid = dd_drift_allows_local_rebuild(dd, b);
    return MDX_OK;
}

static int dd_repartition(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    const auto t0 = std::chrono::steady_clock::now();
    MDX_TRY(mdx_dd_gather_global(h, false));
    MDX_TRY(dd_partition(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    dd->repartition_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MDX_OK;
}

int mdx_dd_on_stale(mdx_handle* h) {
    MdxRange range_stale("mdx decomposed stale list: local rebuild or repartition");
    MdxDecomp* dd = h->dd;
    // The drift (and SHAKE) of the stale step happened, its force call - which starts by constructing the virtual sites -
    // did not: bring the owned sites up to date before their positions travel (gather or halo), or the peers would
    // evaluate this step's forces against last step's M sites.
    if (h->n_vsites && h->in_slot_space) MDX_TRY(mdx_launch_vsite_construct(h, nullptr, 0));
    bool valid = true;
    MDX_TRY(dd_local_set_still_valid(h, &valid));
    static const bool dbg_stale = [] { const char* e = std::getenv("MDX_DEBUG_STALE"); return e && e[0] == '1'; }();
    if (dbg_stale) {
        float d2; std::memcpy(&d2, &dd->spec_bits, 4);
        std::fprintf(stderr, "[mdx] rank %d: stale list -> %s (largest drift since the partition %.4g A, margin %.3g A, %u local rebuilds since)\n", dd->rank,
                     valid ? "local rebuild" : "repartition", (double)std::sqrt(d2), (double)dd->margin, dd->local_rebuilds_since);
    }
    if (valid) { dd->local_rebuilds++; dd->local_rebuilds_since++; h->list_valid = false; }
    else MDX_TRY(dd_repartition(h));
    MDX_TRY(mdx_rebuild(h));
    if (valid && h->n_vsites && dd->world > 1) {   // same atom set: the ghosts' sites were packed before the construction above
        dd->halo_step = -1;
        MDX_TRY(mdx_dd_halo_begin(h));
        MDX_TRY(mdx_dd_halo_end(h));
    }
    return MDX_OK;
}

// ---- collective read-back -------------------------------------------------------------------------------------------
int mdx_dd_download(mdx_handle* h, int which, float* dst) {
    MdxDecomp* dd = h->dd;
    if (which == MDX_FORCE) MDX_TRY(mdx_ensure_ready(h));
    MDX_TRY(mdx_dd_gather_global(h, which == MDX_FORCE));
    const uint32_t N = h->N;
    std::vector<float4> hb(N);
    const float4* src = which == MDX_POS ? dd->g_pos : (which == MDX_VEL ? dd->g_vel : dd->g_frc);
    HIP_TRY(hipMemcpyAsync(hb.data(), src, sizeof(float4) * N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (uint32_t i = 0; i < N; ++i) {
        float v[3] = {hb[i].x, hb[i].y, hb[i].z};
        if (which == MDX_POS)      // back from the owner's frame into the box
            for (int d = 0; d < 3; ++d) {
                const float lo = dd->box_lo[d], L = dd->box_len[d];
                float t = v[d] - std::floor((v[d] - lo) / L) * L;
                if (t < lo) t += L;
                if (t >= lo + L) t -= L;
                v[d] = t;
            }
        dst[3 * i] = v[0]; dst[3 * i + 1] = v[1]; dst[3 * i + 2] = v[2];
    }
    return MDX_OK;
}

// ---- attach / destroy -------------------------------------------------------------------------------------------------
void mdx_dd_destroy(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    if (!dd) return;
    if (dd->comm_stream && dd->comm_stream != h->stream) (void)hipStreamSynchronize(dd->comm_stream);
    void* ptrs[] = {dd->anchor, dd->g_pos, dd->g_vel, dd->g_frc, dd->cls, dd->owner, dd->shift_code, dd->send_mask, dd->flags, dd->scan,
                    dd->scan_sums, dd->pos_at_part, dd->owned_gid, dd->send_ids,
                    dd->recv_ids, dd->recv_shift, dd->send_buf, dd->recv_buf, dd->frc_send, dd->frc_recv, dd->gat_send, dd->gat_recv, dd->red, dd->drift_bits, dd->pipe_ctl, dd->send_cnt, dd->send_rows};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (dd->ev_packed) (void)hipEventDestroy(dd->ev_packed);
    if (dd->ev_arrived) (void)hipEventDestroy(dd->ev_arrived);
    if (dd->ev_fork) (void)hipEventDestroy(dd->ev_fork);
    if (dd->ev_interior) (void)hipEventDestroy(dd->ev_interior);
    if (dd->side_stream) { (void)hipStreamSynchronize(dd->side_stream); (void)hipStreamDestroy(dd->side_stream); }
    if (dd->comm_stream && dd->comm_stream != h->stream) (void)hipStreamDestroy(dd->comm_stream);
    delete dd->tr;
    delete dd;
    h->dd = nullptr;
}

// bricks from the handle's box; the halo: ghosts are kept out to r_list + margin (+ ext), atoms may then drift margin / 2
// before a rank can miss a neighbour (both again whenever the box changes: the barostat)
static void dd_set_bricks(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    for (int d = 0; d < 3; ++d) {
        dd->box_lo[d] = h->box_lo[d]; dd->box_len[d] = h->box_hi[d] - h->box_lo[d];
        dd->brick_lo[d] = dd->box_lo[d] + dd->box_len[d] * (float)dd->coord[d] / (float)dd->grid[d];
        dd->brick_hi[d] = dd->box_lo[d] + dd->box_len[d] * (float)(dd->coord[d] + 1) / (float)dd->grid[d];
    }
}
static int dd_set_halo(mdx_handle* h) {
    MdxDecomp* dd = h->dd;
    double room = 4.4;
    for (int d = 0; d < 3; ++d)
        if (dd->grid[d] > 1) room = std::min(room, (double)dd->box_len[d] * (1.0 - 1.0 / dd->grid[d]) / 2.0 - dd->r_list - dd->ext - 0.01);
    {
        const char* e = std::getenv("MDX_HALO_MARGIN");
        if (e) room = std::min(room, std::max(0.0, std::atof(e)));
    }
    dd->margin = (float)std::max(0.0, room);
    dd->halo = dd->r_list + dd->margin + dd->ext;
    for (int d = 0; d < 3; ++d)
        if (dd->grid[d] > 1 && dd->box_len[d] / dd->grid[d] + 2.0f * dd->halo > dd->box_len[d] + 1e-3f)
            FAIL(MDX_EPARAM, "decomposition: brick + 2 halo exceeds the box: an atom would be needed under two images (box too small for this many ranks)");
    return MDX_OK;
}

// ---- what the single-GPU workflows do between steps, on a decomposed handle ---------------------------------------------
// Barostat (BarostatCfg, /root/reference src/properties/crystal.rs:312-315): the pressure is all-reduced, so every rank
// computes the same mu; the gathered global coordinates are scaled about box_lo on every rank alike, the box, the bricks and
// the halo follow, and the ranks repartition from that state.
__global__ void dd_scale_global_kernel(uint32_t N, float4* __restrict__ g_pos, DdPart p, float mu) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 q = g_pos[i];
    q.x = p.lo[0] + mu * (dd_wrap1(q.x, p.lo[0], p.len[0]) - p.lo[0]);
    q.y = p.lo[1] + mu * (dd_wrap1(q.y, p.lo[1], p.len[1]) - p.lo[1]);
    q.z = p.lo[2] + mu * (dd_wrap1(q.z, p.lo[2], p.len[2]) - p.lo[2]);
    g_pos[i] = q;
}
int mdx_dd_rescale_box(mdx_handle* h, const float hi[3], float mu) {
    MdxDecomp* dd = h->dd;
    MDX_TRY(mdx_dd_gather_global(h, false));
    hipLaunchKernelGGL(dd_scale_global_kernel, dim3(div_up(h->N, 256)), dim3(256), 0, h->stream, h->N, dd->g_pos, make_part(dd), mu);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int d = 0; d < 3; ++d) h->box_hi[d] = hi[d];
    dd_set_bricks(h);
    MDX_TRY(dd_set_halo(h));
    MDX_TRY(mdx_pme_setup(h));          // mesh spacing and theta(m) follow the box
    h->list_valid = false; h->forces_valid = false;
    MDX_TRY(dd_partition(h));
    MDX_TRY(mdx_rebuild(h));
    return MDX_OK;
}
// Host mutation of a joined handle.  The state is distributed, so a write goes through the replicated global arrays: gather what
// the ranks hold, overwrite the rows the caller names (every rank is handed the same rows: the call is collective, like
// mdx_step with external forces), and re-derive owners, ghosts and halo lists from the result - a repartition, ~1 ms at 1 M
// atoms.  Correct for every caller the reference has (box packing moves atoms between MD phases, the docking loop between
// single points); a pose loop that wants the single-GPU latency keeps its handle undecomposed.
int mdx_dd_upload(mdx_handle* h, int which, uint32_t first, uint32_t count, const float4* host_rows) {
    MdxDecomp* dd = h->dd;
    MDX_TRY(mdx_dd_gather_global(h, false));
    float4* dst = (which == MDX_POS ? dd->g_pos : dd->g_vel) + first;
    HIP_TRY(hipMemcpyAsync(dst, host_rows, sizeof(float4) * (size_t)count, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->list_valid = false; h->forces_valid = false;
    MDX_TRY(dd_partition(h));
    MDX_TRY(mdx_rebuild(h));
    h->moved_outside = true; h->prune_pending = true;
    return MDX_OK;
}

__global__ void dd_scale_about_kernel(uint32_t N, float4* __restrict__ g_pos, float cx, float cy, float cz, float mx, float my, float mz) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 q = g_pos[i];      // (an affine map commutes with the periodic images: an atom held in its owner's frame stays consistent)
    q.x = cx + mx * (q.x - cx); q.y = cy + my * (q.y - cy); q.z = cz + mz * (q.z - cz);
    g_pos[i] = q;
}
// SimBox::new + rebuild on every rank alike; with centre / mu the gathered coordinates follow affinely first (shrink_cell_towards)
int mdx_dd_set_box(mdx_handle* h, const float lo[3], const float hi[3], const float* centre_or_null, const float* mu_or_null) {
    MdxDecomp* dd = h->dd;
    MDX_TRY(mdx_dd_gather_global(h, false));
    if (centre_or_null && mu_or_null) {
        hipLaunchKernelGGL(dd_scale_about_kernel, dim3(div_up(h->N, 256)), dim3(256), 0, h->stream, h->N, dd->g_pos, centre_or_null[0],
                           centre_or_null[1], centre_or_null[2], mu_or_null[0], mu_or_null[1], mu_or_null[2]);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int d = 0; d < 3; ++d) { h->box_lo[d] = lo[d]; h->box_hi[d] = hi[d]; }
    dd_set_bricks(h);
    MDX_TRY(dd_set_halo(h));
    MDX_TRY(mdx_pme_setup(h));          // mesh spacing and theta(m) follow the box
    h->list_valid = false; h->forces_valid = false;
    MDX_TRY(dd_partition(h));
    MDX_TRY(mdx_rebuild(h));
    h->moved_outside = true; h->prune_pending = true;
    return MDX_OK;
}

// Minimiser: the accepted state is kept as a copy of the gathered global arrays; a refused move goes back to it.
int mdx_dd_save_global(mdx_handle* h, float4* backup /* [N] device */) {
    MDX_TRY(mdx_dd_gather_global(h, false));
    HIP_TRY(hipMemcpyAsync(backup, h->dd->g_pos, sizeof(float4) * (size_t)h->N, hipMemcpyDeviceToDevice, h->stream));
    return MDX_OK;
}
int mdx_dd_restore_global(mdx_handle* h, const float4* backup) {
    HIP_TRY(hipMemcpyAsync(h->dd->g_pos, backup, sizeof(float4) * (size_t)h->N, hipMemcpyDeviceToDevice, h->stream));
    h->list_valid = false; h->forces_valid = false;
    MDX_TRY(dd_partition(h));
    MDX_TRY(mdx_rebuild(h));
    return MDX_OK;
}

// One send/recv group of a typical halo message on this transport, in microseconds (the largest any rank saw): 8192 rows of 16 B to
// and from every peer, three groups to warm the path up, eight timed between two events on `st`.  Collective.
static int dd_measure_wire_us(MdxTransport* tr, hipStream_t st, float* us_out) {
    *us_out = 0.f;
    if (tr->world <= 1) return MDX_OK;
    const uint32_t rows = 8192u, n = rows * (uint32_t)(tr->world - 1);
    float4 *snd = nullptr, *rcv = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto done = [&](int rc) {
        if (snd) (void)hipFree(snd);
        if (rcv) (void)hipFree(rcv);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        return rc;
    };
    if (hipMalloc((void**)&snd, sizeof(float4) * n) != hipSuccess || hipMalloc((void**)&rcv, sizeof(float4) * n) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipMemsetAsync(snd, 0, sizeof(float4) * n, st) != hipSuccess) {
        mdx_set_error("decomposition: no memory for the wire-time probe");
        return done(MDX_EOOM);
    }
    std::vector<MdxSeg> segs;
    uint32_t r0 = 0;
    for (int q = 0; q < tr->world; ++q) if (q != tr->rank) { segs.push_back({q, r0, rows}); r0 += rows; }
    for (int k = 0; k < 3; ++k) { const int rc = tr->exchange(snd, segs, rcv, segs, st); if (rc != MDX_OK) return done(rc); }
    if (hipEventRecord(e0, st) != hipSuccess) return done(MDX_EDEVICE);
    for (int k = 0; k < 8; ++k) { const int rc = tr->exchange(snd, segs, rcv, segs, st); if (rc != MDX_OK) return done(rc); }
    float ms = 0.f;
    if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
        mdx_set_error("decomposition: the wire-time probe failed");
        return done(MDX_EDEVICE);
    }
    float us = ms * 1000.f / 8.f;
    uint32_t words[DD_MAX_WORLD], mine;
    std::memcpy(&mine, &us, 4);
    if (tr->all_gather_u32(mine, words, st) != MDX_OK) return done(MDX_EDEVICE);
    for (int q = 0; q < tr->world; ++q) { float v; std::memcpy(&v, &words[q], 4); if (v > us) us = v; }
    *us_out = us;
    return done(MDX_OK);
}

int mdx_dd_attach(mdx_handle* h, MdxTransport* tr) {
    auto bail = [&](int rc) { std::string keep = mdx_last_error(); if (!h->dd) delete tr; else mdx_dd_destroy(h); mdx_set_error(keep); return rc; };
    if (!(h->per[0] && h->per[1] && h->per[2])) { mdx_set_error("spatial decomposition needs a fully periodic box"); return bail(MDX_EPARAM); }
    if (h->n_local != h->N) { mdx_set_error("the handle already simulates a subset"); return bail(MDX_EPARAM); }
    if (hipSetDevice(h->device) != hipSuccess) { mdx_set_error("hipSetDevice failed"); return bail(MDX_EDEVICE); }
    const uint32_t N = h->N;
    MdxDecomp* dd = new MdxDecomp();
    dd->tr = tr; dd->rank = tr->rank; dd->world = tr->world;
    h->dd = dd;
    if (dd->world > DD_MAX_WORLD) { mdx_set_error("decomposition: more than 32 ranks"); return bail(MDX_EPARAM); }
    process_grid(dd->world, dd->grid);
    dd->coord[0] = dd->rank / (dd->grid[1] * dd->grid[2]); dd->coord[1] = (dd->rank / dd->grid[2]) % dd->grid[1]; dd->coord[2] = dd->rank % dd->grid[2];
    dd_set_bricks(h);
    dd->r_list = h->r_list;
    if (std::isinf(dd->r_list)) { mdx_set_error("spatial decomposition needs finite cut-offs"); return bail(MDX_EPARAM); }
    // ownership anchors and the reach of a constraint cluster / virtual-site family from its anchor
    std::vector<uint32_t> anchor(N);
    for (uint32_t i = 0; i < N; ++i) anchor[i] = i;
    double ext = 0.0;
    for (const auto& g : h->h_groups) {
        double dist[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) dist[a][b] = a == b ? 0.0 : 1e30;
        for (uint32_t c = 0; c < g.ncons; ++c) { dist[g.ca[c]][g.cb[c]] = std::min(dist[g.ca[c]][g.cb[c]], (double)g.len[c]); dist[g.cb[c]][g.ca[c]] = dist[g.ca[c]][g.cb[c]]; }
        for (int k = 0; k < 4; ++k) for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) dist[a][b] = std::min(dist[a][b], dist[a][k] + dist[k][b]);
        for (uint32_t k = 0; k < g.natoms; ++k) { anchor[g.atom[k]] = g.atom[0]; if (dist[0][k] < 1e29) ext = std::max(ext, dist[0][k]); }
    }
    for (const auto& st : h->h_star5)      // X-H4: the centre anchors its hydrogens
        for (int k = 1; k < 5; ++k) { anchor[st.atom[k]] = st.atom[0]; ext = std::max(ext, (double)st.len[k - 1]); }
    for (const auto& v : h->h_vsites) {   // a site follows its first parent (inside the parents' triangle: no extra reach)
        const uint32_t a = anchor[v.p0];
        anchor[v.site] = a; anchor[v.p1] = a; anchor[v.p2] = a;
        if (!h->vsites_convex) ext = std::max(ext, 3.0);
    }
    if (!h->h_vsites.empty() && h->h_groups.empty()) ext = std::max(ext, 2.0);   // flexible parents: a bond length of room
    dd->ext = (float)(ext * 1.05 + (ext > 0.0 ? 0.05 : 0.0));
    if (dd_set_halo(h) != MDX_OK) return bail(MDX_EPARAM);
    {   // Half shell (every cross-rank pair on one rank, ghost forces travel back: TWO messages per step) or full shell (cross pairs
        // on both ranks, twice the ghosts, ONE message)?  Measured round 6 on rank 0 of the 1 M-atom box with a stated time per
        // message (profiles/r06_one_rank_of_N.txt; step in ms, half / full): 8 ranks 0.132 / 0.141 at 0 us, 0.149 / 0.142 at 5,
        // 0.162 / 0.148 at 10, 0.208 / 0.173 at 25, 0.272 / 0.196 at 50; 4 ranks 0.200 / 0.219 at 0, 0.259 / 0.249 at 25; 2 ranks
        // 0.330 / 0.335 at 0, 0.395 / 0.365 at 25 - a microsecond of message time costs the half shell 2.4-3.0 us of step, the full
        // shell 1.2-1.3: they cross at ~4 us per message (16 us on four ranks) in that session and at ~10 us in a second one (the half
        // shell's step varies more from box to box: 0.122-0.132 ms at 0 us, 0.184-0.208 at 25; the full shell's does not).  The crossing
        // is taken at 8 us (16 on four ranks).  So a transport whose message time means something
        // is measured here, once (every rank takes the largest value any rank saw: the choice is part of the partition), and the
        // half shell stays only below that crossing.  Half shell needs Newton's third law across the rank boundary: the half-list
        // pair kernel.  MDX_HALF_SHELL=0 / 1 pins the choice; MDX_HALF_SHELL_WIRE_US moves the crossing.
        const char* e = std::getenv("MDX_HALF_SHELL");
        bool hs = dd->world > 1 && mdx_nb_half(h);
        if (e && (e[0] == '0' || e[0] == '1')) hs = hs && e[0] == '1';
        else if (hs && tr->wire_time_decides()) {
            if (dd_measure_wire_us(tr, h->stream, &dd->wire_us) != MDX_OK) return bail(MDX_EDEVICE);
            const char* t = std::getenv("MDX_HALF_SHELL_WIRE_US");
            const float crossing = t ? (float)std::atof(t) : (dd->world == 4 ? 16.f : 8.f);
            hs = dd->wire_us < crossing;
        }
        dd->half_shell = hs;
    }
    {
        const char* e = std::getenv("MDX_HALO_OVERLAP");
        dd->overlap = !(e && e[0] == '0');
        dd->tune_phase = (e && (e[0] == '0' || e[0] == '1')) ? 2 : 0;
    }
    int rc = MDX_OK;
#define DD_TRY(x) do { rc = (x); if (rc != MDX_OK) return bail(rc); } while (0)
#define DD_HIP(x) do { if ((x) != hipSuccess) { mdx_set_error(#x " failed"); return bail(MDX_EDEVICE); } } while (0)
    DD_TRY(dd_alloc(&dd->anchor, N)); DD_TRY(dd_alloc(&dd->g_pos, N)); DD_TRY(dd_alloc(&dd->g_vel, N));
    DD_TRY(dd_alloc(&dd->cls, N)); DD_TRY(dd_alloc(&dd->owner, N)); DD_TRY(dd_alloc(&dd->shift_code, N)); DD_TRY(dd_alloc(&dd->send_mask, N));
    DD_TRY(dd_alloc(&dd->pos_at_part, N)); DD_TRY(dd_alloc(&dd->owned_gid, N)); DD_TRY(dd_alloc(&dd->red, 64));
    dd->cap_local = N;
    // Every RCCL operation of a handle goes to ONE stream.  By default that is the compute stream itself: pack -> send/recv
    // group -> unpack run back to back without cross-stream event hops (each hop is ~5-10 us of idle time on this stack; the
    // overlap with the message comes from the interior tiles on the side stream).  MDX_COMM_STREAM=1: a separate stream.
    DD_TRY(mdx_stream_unmask(h));      // (a joined handle's chain has the mesh all-reduce inside: no CU split)
    {
        const char* e = std::getenv("MDX_COMM_STREAM");
        if (e && e[0] == '1') DD_HIP(hipStreamCreateWithFlags(&dd->comm_stream, hipStreamNonBlocking));
        else dd->comm_stream = h->stream;
    }
    DD_HIP(hipEventCreateWithFlags(&dd->ev_packed, hipEventDisableTiming));
    DD_HIP(hipEventCreateWithFlags(&dd->ev_arrived, hipEventDisableTiming));
    {   // the side stream (interior tiles) runs at the lowest priority, so that the kernels on the compute stream - the ones the messages wait
        // for - are dispatched first when both have workgroups pending (rank 0 of 8: split step 0.154 -> 0.142 ms, 0.206 -> 0.186 with 25 us
        // per message; MDX_SIDE_PRIO=0: A/B)
        const char* e = std::getenv("MDX_SIDE_PRIO");
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (!(e && e[0] == '0')) DD_HIP(hipStreamCreateWithPriority(&dd->side_stream, hipStreamNonBlocking, lo));
        else DD_HIP(hipStreamCreateWithFlags(&dd->side_stream, hipStreamNonBlocking));
    }
    DD_HIP(hipEventCreateWithFlags(&dd->ev_fork, hipEventDisableTiming));
    DD_HIP(hipEventCreateWithFlags(&dd->ev_interior, hipEventDisableTiming));
    DD_HIP(hipMemcpyAsync(dd->anchor, anchor.data(), sizeof(uint32_t) * N, hipMemcpyHostToDevice, h->stream));
    DD_HIP(hipStreamSynchronize(h->stream));
    // the global state every rank starts from is the handle's own (it was created from the whole system)
    DD_TRY(mdx_unsort_state(h));
    DD_HIP(hipMemcpyAsync(dd->g_pos, h->d.pos_orig, sizeof(float4) * N, hipMemcpyDeviceToDevice, h->stream));
    DD_HIP(hipMemcpyAsync(dd->g_vel, h->d.vel_orig, sizeof(float4) * N, hipMemcpyDeviceToDevice, h->stream));
    {   // the drift pass of a decomposed step carries the halo pack and the add of the returned ghost forces (MDX_HALO_FOLD=0: A/B)
        const char* f = std::getenv("MDX_HALO_FOLD");
        // (round 6: with the full shell the pass would only pack - no ghost forces come back; built, agrees (fuzz), measured SLOWER than the
        // pack kernel it saves - rank 0 of 8: 0.140 against 0.135 ms at 0 us per message, 0.171 against 0.163 at 25 - so it stays behind
        // MDX_HALO_FOLD_FULL_SHELL=1)
        static const bool fold_full = [] { const char* x = std::getenv("MDX_HALO_FOLD_FULL_SHELL"); return x && x[0] == '1'; }();
        dd->fold_ok = dd->world > 1 && (dd->half_shell || fold_full) && dd->comm_stream == h->stream && !(f && f[0] == '0');
    }
    h->want_tile_split = dd->world > 1 && (dd->overlap || dd->tune_phase < 2);
    if (h->pme_on) {   // the reciprocal-space chain of a decomposed handle runs on the handle's own stream (mesh all-reduce inside)
        if (h->pme_overlap && h->stream_pme) DD_HIP(hipStreamSynchronize(h->stream_pme));
        h->pme_overlap = false;
        DD_TRY(mdx_pme_setup(h));
    }
    {   // One Verlet skin for all ranks.  A handle created with mdx_config.skin == 0 tunes its skin from its own wall clock while it
        // steps alone (mdx_step); handles that stepped before joining may arrive here with different skins - different list radii,
        // halo widths and stale thresholds, i.e. send lists that disagree with the peers' receive lists.  The tuning stops here
        // (ranks' clocks do not agree; mdx_get_skin reports tuning = 0 from now on) and every rank takes the smallest skin any
        // rank holds (a legal choice everywhere: it was checked against the same box).  Placed behind every check a rank can fail
        // on its own: a rank that refuses to join must not leave its peers waiting in a collective.
        uint32_t words[DD_MAX_WORLD];
        uint32_t mine; std::memcpy(&mine, &h->cfg.skin, 4);
        if (tr->all_gather_u32(mine, words, h->stream) != MDX_OK) return bail(MDX_EDEVICE);
        float smin = h->cfg.skin;
        for (int q = 0; q < tr->world; ++q) { float s; std::memcpy(&s, &words[q], 4); if (s >= 0.f && s < smin) smin = s; }
        h->skin_tune.phase = 4;
        if (smin != h->cfg.skin) {
            h->cfg.skin = smin;
            h->r_list = std::max(h->cfg.lj_cutoff, h->cfg.coulomb_cutoff) + smin;
            h->list_valid = false; h->stretch_samples = 0;
            h->inner_skin_auto = 0.f; h->dual_auto_off = false; h->dual_win_steps = 0; h->dual_win_prunes = 0;
            dd->r_list = h->r_list;
            if (dd_set_halo(h) != MDX_OK) return bail(MDX_EPARAM);
        }
    }
    DD_TRY(dd_partition(h));
    DD_TRY(mdx_rebuild(h));
    h->forces_valid = false;
    // Every enqueued step of a decomposed handle issues a halo group and a force-return group, so all ranks must cut their chunks
    // alike.  The chunk-length predictor of mdx_step works from the handle's own history of rebuild-free stretches: a rank that
    // stepped before it joined (equilibration on the handle it later decomposes) would cut differently from its peers.  From here on
    // every rank sees the same stale steps (the flag rides on the messages), so the statistics start afresh and stay in step.
    h->steps_since_rebuild = 0; h->stretch_samples = 0; h->stretch_mean = 0.f; h->stretch_dev = 0.f;
    h->dual_win_steps = 0; h->dual_win_prunes = 0;
#undef DD_TRY
#undef DD_HIP
    return MDX_OK;
}

// ---- C ABI: what a host may ask about the decomposition -------------------------------------------------------------
extern "C" int mdx_comm_info(const mdx_handle* h, int* rank, int* world, int grid[3], uint32_t* n_owned, uint32_t* n_ghost, float* halo) {
    if (!h) FAIL(MDX_EPARAM, "null handle");
    const MdxDecomp* dd = h->dd;
    if (rank) *rank = dd ? dd->rank : 0;
    if (world) *world = dd ? dd->world : 1;
    if (grid) for (int d = 0; d < 3; ++d) grid[d] = dd ? dd->grid[d] : 1;
    if (n_owned) *n_owned = dd ? dd->n_owned : h->N;
    if (n_ghost) *n_ghost = dd ? dd->n_local - dd->n_owned : 0;
    if (halo) *halo = dd ? dd->halo : 0.f;
    return MDX_OK;
}

extern "C" int mdx_comm_diag_read(mdx_handle* h, mdx_comm_diag* out) {
    if (!h || !out) FAIL(MDX_EPARAM, "null argument");
    if (!h->dd) FAIL(MDX_EPARAM, "the handle has not joined a communicator");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    mdx_prof_collect(h);
    const MdxDecomp* dd = h->dd;
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->transport, sizeof(out->transport), "%s", dd->tr->name());
    out->rank = dd->rank; out->world = dd->world;
    for (int d = 0; d < 3; ++d) out->grid[d] = dd->grid[d];
    dd->tr->wire_info(&out->rccl_version, &out->rccl_comm_count);
    out->half_shell = mdx_dd_half_shell(h) ? 1 : 0;
    out->wire_ns_measured = dd->wire_us < 0.f ? -1 : (int32_t)std::min(2.0e9f, dd->wire_us * 1000.f);
    out->overlap_split = dd->world > 1 ? (dd->tune_phase < 2 ? -1 : (dd->overlap ? 1 : 0)) : 0;
    out->comm_stream_separate = dd->comm_stream != h->stream ? 1 : 0;
    out->n_owned = dd->n_owned; out->n_ghost = dd->n_local - dd->n_owned;
    out->n_tiles = h->T; out->n_interior_tiles = h->tile_split ? h->n_interior : 0;
    out->halo_rows_out = dd->n_send; out->halo_rows_in = dd->n_recv;
    out->halo_bytes_per_step = (uint64_t)(dd->n_send + dd->n_recv) * sizeof(float4) * (mdx_dd_half_shell(h) ? 2u : 1u);
    out->repartitions = dd->repartitions; out->local_rebuilds = dd->local_rebuilds; out->repartition_ms_sum = dd->repartition_ms;
    for (int k = 0; k < MDX_DIAG_PHASES; ++k) { out->phase_ms[k] = dd->phase_ms[k]; out->phase_n[k] = dd->phase_n[k]; }
    return MDX_OK;
}

extern "C" int mdx_comm_debug_partition(mdx_handle* h, uint8_t* cls, uint8_t* owner, uint8_t* image_code, uint32_t* send_mask,
                                        uint32_t* n_send, uint32_t* n_recv, uint32_t* send_ids, uint32_t* recv_ids, uint32_t capacity) {
    if (!h || !h->dd) FAIL(MDX_EPARAM, "the handle has not joined a communicator");
    MdxDecomp* dd = h->dd;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const uint32_t N = h->N;
    if (cls) HIP_TRY(hipMemcpy(cls, dd->cls, N, hipMemcpyDeviceToHost));
    if (owner) HIP_TRY(hipMemcpy(owner, dd->owner, N, hipMemcpyDeviceToHost));
    if (image_code) HIP_TRY(hipMemcpy(image_code, dd->shift_code, N, hipMemcpyDeviceToHost));
    if (send_mask) HIP_TRY(hipMemcpy(send_mask, dd->send_mask, sizeof(uint32_t) * N, hipMemcpyDeviceToHost));
    if (n_send) *n_send = dd->n_send;
    if (n_recv) *n_recv = dd->n_recv;
    if (send_ids && dd->n_send) {
        if (capacity < dd->n_send) FAIL(MDX_EPARAM, "capacity below n_send");
        HIP_TRY(hipMemcpy(send_ids, dd->send_ids, sizeof(uint32_t) * dd->n_send, hipMemcpyDeviceToHost));
    }
    if (recv_ids && dd->n_recv) {
        if (capacity < dd->n_recv) FAIL(MDX_EPARAM, "capacity below n_recv");
        HIP_TRY(hipMemcpy(recv_ids, dd->recv_ids, sizeof(uint32_t) * dd->n_recv, hipMemcpyDeviceToHost));
    }
    return MDX_OK;
}

// Exercises every transport entry point of a joined handle on the wire it really has - including, with one rank, RCCL's
// send/recv to self inside a group - and checks the results: what a single-GPU box can verify of the RCCL leg.
__global__ void dd_selftest_fill_kernel(uint32_t n, float4* a, float seed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = make_float4(seed + (float)i, 2.f * (float)i, -(float)i, 1.f);
}
extern "C" int mdx_comm_selftest_fault(mdx_handle* h) {
    if (!h || !h->dd) FAIL(MDX_EPARAM, "the handle has not joined a communicator");
    MdxDecomp* dd = h->dd;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = dd->comm_stream;
    HIP_TRY(hipStreamSynchronize(h->stream));
    float4 *a = nullptr;
    HIP_TRY(hipMalloc((void**)&a, sizeof(float4) * 64));
    HIP_TRY(hipMemsetAsync(a, 0, sizeof(float4) * 64, st));
    std::vector<MdxSeg> ss{{dd->world + 3, 0u, 16u}}, rs{{dd->world + 3, 32u, 16u}};   // a peer that does not exist
    if (!dd->tr->delivers()) { (void)hipFree(a); return MDX_OK; }     // (the null transport has no wire to fail on)
    // Not collective: every transport refuses a peer outside the communicator BEFORE it meets the others (mdx_comm.hip: the
    // fabric and shared-memory transports check their segments ahead of their barrier), so no rank waits for one that never comes
    const int rc1 = dd->tr->exchange(a, ss, a, rs, st);
    const std::string first = mdx_last_error();
    // the group must be closed and the transport must refuse what follows instead of queueing it
    std::vector<MdxSeg> ok_s{{dd->rank, 0u, 16u}}, ok_r{{dd->rank, 32u, 16u}};
    const int rc2 = dd->tr->exchange(a, ok_s, a, ok_r, st);
    const hipError_t sync = hipStreamSynchronize(st);
    (void)hipFree(a);
    if (rc1 == MDX_OK) FAIL(MDX_EDEVICE, "selftest_fault: a send to a non-existent rank was accepted");
    if (rc2 == MDX_OK) FAIL(MDX_EDEVICE, "selftest_fault: the transport kept going after a failed group");
    if (sync != hipSuccess) FAIL(MDX_EDEVICE, "selftest_fault: the communication stream did not drain");
    mdx_set_error("selftest_fault: reported as expected: " + first);
    return MDX_OK;
}

extern "C" int mdx_comm_selftest(mdx_handle* h) {
    if (!h || !h->dd) FAIL(MDX_EPARAM, "the handle has not joined a communicator");
    MdxDecomp* dd = h->dd;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = dd->comm_stream;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const uint32_t n = 1000;
    float4 *a = nullptr, *b = nullptr;
    HIP_TRY(hipMalloc((void**)&a, sizeof(float4) * n * (size_t)dd->world));
    HIP_TRY(hipMalloc((void**)&b, sizeof(float4) * n * (size_t)dd->world));
    auto done = [&](int rc) { (void)hipFree(a); (void)hipFree(b); return rc; };
    hipLaunchKernelGGL(dd_selftest_fill_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, n, a, 1000.f * (float)dd->rank);
    HIP_TRY(hipMemsetAsync(b, 0, sizeof(float4) * n * (size_t)dd->world, st));
    // every rank sends its block to every rank (itself included) and files what arrives by sender
    std::vector<MdxSeg> ss, rs;
    for (int q = 0; q < dd->world; ++q) { ss.push_back({q, 0u, n}); rs.push_back({q, (uint32_t)q * n, n}); }
    int rc = dd->tr->exchange(a, ss, b, rs, st);
    if (rc != MDX_OK) return done(rc);
    std::vector<float4> hb((size_t)n * dd->world);
    if (hipMemcpyAsync(hb.data(), b, sizeof(float4) * hb.size(), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        { mdx_set_error("selftest: copy failed"); return done(MDX_EDEVICE); }
    if (dd->tr->delivers())
        for (int q = 0; q < dd->world; ++q)
            for (uint32_t i = 0; i < n; i += 97)
                if (hb[(size_t)q * n + i].x != 1000.f * (float)q + (float)i || hb[(size_t)q * n + i].z != -(float)i)
                    { mdx_set_error("selftest: a send/recv segment arrived wrong"); return done(MDX_EDEVICE); }
    // small all-reduces (sum of doubles, max of words), the large f32 sum, the word all-gather
    double v[3] = {1.0, (double)dd->rank, 0.5};
    rc = mdx_dd_allreduce_host(h, v, 3);
    if (rc != MDX_OK) return done(rc);
    const double W = dd->world;
    if (dd->tr->delivers() && dd->world > 1 && (v[0] != W || v[1] != W * (W - 1) / 2 || v[2] != 0.5 * W)) { mdx_set_error("selftest: all-reduce(sum) wrong"); return done(MDX_EDEVICE); }
    double m[1] = {(double)(7 + dd->rank)};
    rc = mdx_dd_allreduce_host(h, m, 1, true);
    if (rc != MDX_OK) return done(rc);
    if (dd->tr->delivers() && m[0] != 7.0 + (W - 1)) { mdx_set_error("selftest: all-reduce(max) wrong"); return done(MDX_EDEVICE); }
    hipLaunchKernelGGL(dd_selftest_fill_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, n, a, 1.f);
    rc = dd->tr->all_reduce_f32((float*)a, 4 * (size_t)n, st);
    if (rc != MDX_OK) return done(rc);
    if (hipMemcpyAsync(hb.data(), a, sizeof(float4) * n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        { mdx_set_error("selftest: copy failed"); return done(MDX_EDEVICE); }
    if (dd->tr->delivers() && (hb[10].x != (float)(W * 11.0) || hb[10].w != (float)W)) { mdx_set_error("selftest: large f32 all-reduce wrong"); return done(MDX_EDEVICE); }
    uint32_t words[DD_MAX_WORLD];
    rc = dd->tr->all_gather_u32(100u + (uint32_t)dd->rank, words, st);
    if (rc != MDX_OK) return done(rc);
    if (dd->tr->delivers()) for (int q = 0; q < dd->world; ++q) if (words[q] != 100u + (uint32_t)q) { mdx_set_error("selftest: all-gather wrong"); return done(MDX_EDEVICE); }
    return done(MDX_OK);
}
