// mdx_internal.h — shared declarations of the gfx950 MD engine (not part of the public ABI).
//
// Data layout in HBM ("slot space"):
//   Atoms are re-ordered at every neighbour rebuild into TILES of 64 consecutive slots (one
//   wavefront = one tile, one atom per lane); a tile is 8 CLUSTERS of 8 slots.  Tiles are bricks
//   of an x-y column grid, z-sorted inside a column, and sub-sorted so every cluster is compact.
//   Unused slots hold dummy atoms (charge 0, eps 0, infinite mass) parked far away at distinct
//   coordinates.  All per-step kernels work on slot-space float4 arrays (16 B per lane: the
//   widest coalesced access):
//     posq [S] x,y,z, q*sqrt(k_e)          lj   [S] (sigma/2 | sqrt(sigma), sqrt(24 eps))
//     vel  [S] vx,vy,vz, 418.4/mass (0 = never integrated)
//     force[S] fx,fy,fz,-                  ref  [S] position at the last rebuild, .w = path length since the last
//                                                   pruning pass of the dual list
//   The pair list is per tile: a run of (j-cluster, image shift, i-cluster mask) entries, the
//   ones that need exclusion masks first, each run padded to a multiple of 8 entries (= one
//   64-atom chunk staged through LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include <string>
#include <vector>
#include "../../include/mdx.h"

#define MDX_TILE 64
#define MDX_CLUSTER 8
#define MDX_CL_PER_TILE 8
#define MDX_ACC_CONV 418.4f
#define MDX_KB 0.0019872041
#define MDX_INVALID 0xFFFFFFFFu
#define MDX_DUMMY_BASE 1.0e6f
#define MDX_DUMMY_STEP 64.0f
#define MDX_MAX_CHUNK 64
#define MDX_MAX_GROUPS 255  // mdx_set_energy_groups: a group index is a byte

#define MDX_ESTRIDE 5   // {lj, coulomb, virial, cross (alchemical: unscaled energy of the coupled pairs), dU/dlambda} per slot
#define MDX_EPART 256   // the pair kernel spreads its energy atomics over this many slots (contended f64 atomics cost ~10 ns each)
enum { EN_BOND = 0, EN_ANGLE, EN_DIHEDRAL, EN_LJ, EN_COUL, EN_LJ14, EN_COUL14, EN_KIN, EN_RECIP, EN_VIRIAL, EN_COUNT };

struct GridParams {
    float lo[3];       // origin of the column grid (box_lo, or bounding box in vacuum)
    float len[3];      // box edge (periodic) or bounding extent
    float inv_col[2];  // 1 / column edge in x, y
    float inv_zbin;    // 1 / fine z-bin height
    int ncx, ncy, nzb; // columns in x, y; z-bins per column
    int npop;          // 1, or 2 on a half-shell decomposed handle: owned atoms and ghost atoms are binned into SEPARATE column
                       // sets (column = (pop * ncx + cx) * ncy + cy), so every tile and cluster is all-owned or all-ghost and
                       // "no ghost-ghost pair is evaluated" is a cluster-level decision (mdx_decomp.hip: forces on ghosts go back)
    int per[3];        // periodic per dimension (a decomposed dimension is not periodic locally)
    // Piecewise column grid (x, y of a half-shell decomposed handle, dimension by dimension; pw[d] = 0: uniform).  The local region
    // of a rank is halo | brick | halo; one uniform grid over it cuts the columns at the brick faces into an owned part and a ghost
    // part - slivers a few Angstrom wide whose 64-atom tiles are 30-110 A tall (needle clusters: several times the neighbours of a
    // compact one, and lists that overflowed the list build's LDS buffer).  Here each of the three regions has whole columns of
    // its own width: nh of them per halo region, nbk over the brick.
    int pw[2], nh[2], nbk[2];
    float b_lo[2], b_hi[2], inv_wh[2], inv_wb[2];
};
// column of a coordinate (unclamped) and lower edge of a column, for either grid form
__host__ __device__ static inline int mdx_col_of(const GridParams& g, int d, float x) {
    if (!g.pw[d]) return (int)floorf((x - g.lo[d]) * g.inv_col[d]);
    if (x < g.b_lo[d]) return (int)floorf((x - g.lo[d]) * g.inv_wh[d]);
    if (x < g.b_hi[d]) { const int i = (int)((x - g.b_lo[d]) * g.inv_wb[d]); return g.nh[d] + (i < g.nbk[d] ? i : g.nbk[d] - 1); }
    return g.nh[d] + g.nbk[d] + (int)floorf((x - g.b_hi[d]) * g.inv_wh[d]);
}
__host__ __device__ static inline float mdx_col_lo(const GridParams& g, int d, int i) {
    if (!g.pw[d]) return g.lo[d] + (float)i / g.inv_col[d];
    if (i < g.nh[d]) return g.lo[d] + (float)i / g.inv_wh[d];
    if (i == g.nh[d]) return g.b_lo[d];
    if (i < g.nh[d] + g.nbk[d]) return g.b_lo[d] + (float)(i - g.nh[d]) / g.inv_wb[d];
    if (i == g.nh[d] + g.nbk[d]) return g.b_hi[d];
    return g.b_hi[d] + (float)(i - g.nh[d] - g.nbk[d]) / g.inv_wh[d];
}

struct NbParams {
    float rc2_lj, rc2_coul;  // squared cut-offs (FLT_MAX = none)
    float shift[3];          // box edges for image shifts (0 in vacuum)
    float coul_shift;        // 1/rc  (shifted potential), or c_rf (reaction field)
    float k_rf2;             // 2*k_rf (reaction field force term)
    float k_rf;
    float alpha;             // Ewald real-space
    float soft2;
    int coul_mode;
    float alch_scale;        // 1 - lambda: factor on pairs with exactly one atom in the coupled molecule (ALCH kernels)
    float sc_al, sc_alpha, sc_sigmin;   // soft core of those pairs: r_sc^6 = sc_al sigma^6 + r^6, sc_al = alpha lambda (0: linear coupling)
    int geometric;           // combining rule
    int lj_on, coul_on;
    // Ewald real space, force flavour: the smooth part g(r^2) = [erf(beta r)/r - 2 beta/sqrt(pi) exp(-beta^2 r^2)] / r^2 from a table
    // (mdx_pair_dev.h: EWALD_TAB_*): F/r = q q (1/r^3 - g) without v_exp / v_rcp in the pair loop.  null: the closed form.
    const float4* etab; uint32_t etab_n;
    float etab_scale; uint32_t etab_shift;      // x = r^2 * etab_scale + EWALD_TAB_C; index = (bits(x) >> etab_shift) - (bits(EWALD_TAB_C) >> etab_shift)
};

struct BondedParams {
    float box[3];  // edges, 0 when not periodic
    float inv_box[3];
    float scale14_lj, scale14_coul;
    int geometric, lj_on, coul_on;
    float ewald_beta;    // ROLE_EWALD_EXCL: remove erf(beta r)/r of pairs the real-space sum skips
    int skip_bonded;     // MdOverrides.bonded_disabled: keep only the Ewald exclusion corrections
};

// Control block in device memory: the rebuild trigger.  disp2[s] holds, as the bit pattern of a
// non-negative float, the largest squared displacement (w.r.t. the positions of the last rebuild)
// seen after the drift of chunk-step s-1; every kernel of chunk-step s first tests
// disp2[s] > thr and turns into a no-op when the list has gone stale, so the host may enqueue a
// whole chunk of steps without synchronising.
//
// Dual pair list: prune[s] != 0 <=> after the drift of chunk-step s-1 some atom's path length since the last
// pruning pass exceeded inner_skin/2; the pair kernel of chunk-step s-1's force call then walks the OUTER masks,
// re-derives the inner masks and clears the path accumulators (ref[].w).  Device-side only, no host decision.
struct StepCtl {
    uint32_t disp2[MDX_MAX_CHUNK + 2];
    uint32_t nonfinite;
    uint32_t pad;
    uint32_t prune[MDX_MAX_CHUNK + 2];
    // decomposed handle with interior / boundary split: raised by the halo unpack when a GHOST's path length since the last
    // pruning pass of the boundary tiles exceeded inner_skin/2 (interior tiles never see a ghost and never read it)
    uint32_t prune_ghost[MDX_MAX_CHUNK + 2];
    // one launch per step (mdx_pair_dev.h NbArgs::st_*): the kick of step s exceeded what the words of step s - 1 had granted it
    uint32_t viol[MDX_MAX_CHUNK + 2];
};

// One launch per step (mdx_step "onepass"; mdx_pair_dev.h NbArgs::st_*)
struct OnePassNow {      // what mdx_launch_nonbonded adds to the argument block of the launch being enqueued (null fprev: nothing)
    const float4* yin = nullptr; const float4* fprev = nullptr; float4* fcur = nullptr; float4* yout = nullptr; float4* fnext = nullptr;
    float dt = 0.f; bool last = false, first = false; uint32_t* disp_out = nullptr; uint32_t* prune_out = nullptr; uint32_t* viol = nullptr;
};

// Decomposed handle, steps of the fused bonded + kick + drift pass: the pass also packs the halo and adds the ghost forces the peers
// returned for the previous force call (mdx_integrate.hip, mdx_decomp.hip).  Its workgroups count themselves in here - two levels, by
// blockIdx & 7 and then by shard - so that the one that arrives last can write the flag rows of the message (the "list went stale"
// word is complete only then).
struct __attribute__((aligned(128))) PipeCtl { uint32_t m1_shard[16]; uint32_t rows_overflow; uint32_t pad[15]; };

// One (term, atom-of-that-term) record of the atom-owned bonded gather (mdx_bonded.hip).
enum { ROLE_BOND = 0, ROLE_ANGLE = 1, ROLE_DIHEDRAL = 2, ROLE_PAIR14 = 3, ROLE_EWALD_EXCL = 4 };
// 16 bytes: the parameters live in a table of DISTINCT parameter sets (role_prm: a force field has a few hundred; the
// two of a water box sit in L1), which halves the record stream the gather reads - 2.33 roles per water atom.
struct __attribute__((aligned(16))) RoleRec {
    uint32_t p[3];   // the term's other atoms, in term order (caller index in *_o, slot in *_s)
    uint32_t meta;   // kind | role << 4 | parameter-set index << 8   (role = this atom's position in the term)
};
// role_prm[index] (float4): bond: k, r0 | angle: k, theta0 | dihedral: v, phase, n | 1-4: sigma, 4 s eps, s ke qq | Ewald exclusion: ke qq

// A cluster of atoms tied by distance constraints (rigid water: 3 atoms / 3 constraints; X-H3: 4 / 3).
struct __attribute__((aligned(16))) ConsGroup {
    uint32_t atom[4];     // caller index (*_o) or slot (*_s); MDX_INVALID pads
    float    len[6];
    uint8_t  ca[6], cb[6]; // constraint k ties local atoms ca[k], cb[k]
    uint32_t ncons, natoms;
    uint32_t wstep;        // 1: a rigid three-site water whose whole step is water_step_kernel's (mdx_constraints.hip); fills the record's 64 bytes
};
static_assert(sizeof(ConsGroup) == 64, "ConsGroup is one 64-byte record");
// A heavy atom with FOUR constrained hydrogens (ammonium, methane, silane ...: X-H4) - the one cluster shape beyond four atoms that
// "constrain the bonds to hydrogens" (HydrogenConstraint, /root/reference src/ui/panels/md.rs:362-371) produces: a star, atom[0] its
// centre, len[k] the length of the bond to atom[k + 1].  Its own record and its own small kernels: the 64-byte ConsGroup that a
// million rigid waters stream through SETTLE stays as it is.
struct __attribute__((aligned(16))) ConsStar5 { uint32_t atom[5]; float len[4]; uint32_t pad[3]; };
// The virtual site of a constraint cluster whose three parents are all members of the cluster (the M site of a rigid four-site
// water): the position stage of the constraint solver places it (mdx_constraints.hip) - no launch of its own.
struct __attribute__((aligned(16))) GroupSite {
    uint32_t site;         // caller index (*_o) or slot (*_s); MDX_INVALID: none
    uint8_t k0, k1, k2, on; // local indices of the site's parents p0, p1, p2 in ConsGroup::atom
    float a, b;
};
struct __attribute__((aligned(16))) VSite {  // r_site = r0 + a (r1 - r0) + b (r2 - r0)
    uint32_t site, p0, p1, p2;
    float a, b, pad0, pad1;
};

struct ListCounts {  // per tile
    uint32_t n_masked;  // entries in the masked run (multiple of 8)
    uint32_t n_plain;   // entries in the plain run (multiple of 8)
};

// ---- device buffers owned by a handle -----------------------------------------------------------
struct DeviceState {
    // static per-atom data, caller ("orig") order
    float*    o_qs = nullptr;      // q*sqrt(ke) (0 for bonded_only)
    float2*   o_lj = nullptr;
    float*    o_invm = nullptr;    // ACC_CONV/mass or 0
    float*    o_mass = nullptr;
    float*    o_q = nullptr;       // raw charge (1-4 pairs)
    float2*   o_lj_raw = nullptr;  // (sigma, eps) raw (1-4 pairs)
    uint32_t* excl_off = nullptr;  // merged exclusions + 1-4 CSR, orig order
    uint32_t* excl_idx = nullptr;
    // staging in orig order
    float4* pos_orig = nullptr;
    float4* vel_orig = nullptr;
    float4* ext_orig = nullptr;
    // slot space
    float4* posq = nullptr; float2* lj = nullptr; float4* vel = nullptr; float4* force = nullptr;
    float4* ref = nullptr;
    // Path split (round 6; handles without constraints and virtual sites): the path accumulator of the dual pair list lives in its
    // own array instead of ref[].w, and dprune[] holds |x - ref| at the atom's last pruning pass, so the fused bonded + kick + drift pass
    // bounds the displacement since the rebuild by dprune + path (triangle inequality) and reads 8 B per slot where it read and wrote
    // the 16-B ref row
    float* path = nullptr; float* dprune = nullptr;
    float4* force_b = nullptr;    // one launch per step (mdx_pair_dev.h NbArgs::st_*): the force array rotates through three buffers -
    float4* force_c = nullptr;    // read (previous stage, complete), accumulate, zero for the next - and d.force is whichever holds the current one
    float4* posq_alt = nullptr;   // second position buffer: the fused bonded + kick + drift pass reads t, writes t + dt, then the two swap
    uint32_t* orig_of = nullptr;  // [S]  slot -> local atom index (MDX_INVALID for dummies)
    uint32_t* slot_of = nullptr;  // [N]  GLOBAL atom id -> slot (MDX_INVALID when not simulated here)
    uint32_t* gid = nullptr;      // [cap_local] local atom index -> global atom id
    uint8_t*  lflag = nullptr;    // [cap_local] bit0: ghost (halo copy owned by another rank)
    uint8_t*  slot_flags = nullptr; // [S] bit0: real atom, bit1: owned (not a ghost)
    // grid build scratch
    uint32_t* cell_of = nullptr;      // [N]
    uint32_t* cell_count = nullptr;   // [ncells+1]
    uint32_t* cell_start = nullptr;   // [ncells+1]
    uint32_t* cell_cursor = nullptr;  // [ncells]
    uint32_t* sorted_orig = nullptr;  // [N]
    uint32_t* sorted_tmp = nullptr;   // [N] the cells' members as the atomic scatter left them, before the rank sort
    uint32_t* col_tiles = nullptr;    // [ncol+1]
    uint32_t* tile_start = nullptr;   // [ncol+1]
    uint32_t* tile_col = nullptr;     // [T]
    uint32_t* scan_tmp = nullptr;
    // cluster bounding boxes
    float4* cl_lo = nullptr; float4* cl_hi = nullptr;  // [NC]
    uint8_t* cl_kind = nullptr;                        // [NC] bit 0: the cluster holds an atom with a Lennard-Jones well, bit 1: a charged atom
    // pair list
    ListCounts* list_counts = nullptr;  // [T]
    uint32_t* entry_cnt = nullptr;      // [T+1] entries per tile (masked+plain), then scanned
    uint32_t* entry_off = nullptr;      // [T+1]
    uint32_t* mchunk_cnt = nullptr;     // [T+1]
    uint32_t* mchunk_off = nullptr;     // [T+1]
    uint2*    entries = nullptr;        // [E]
    uint2*    entries_in = nullptr;     // [E] dual list: the inner list (masked run in place, plain run compacted per wave)
    uint32_t* list_cursors = nullptr;   // single-pass list build: 64 region cursors (u64: entries | masked chunks << 32), one 128-B line each, then the two totals
    uint32_t* inner_nch = nullptr;      // [T*8] dual list: chunk-loop bound per (tile, wave of the tile)
    unsigned long long* masks = nullptr; // [MC*64]
    // bonded terms as per-atom role lists: caller order (static) and slot order (per rebuild)
    uint32_t* role_off_o = nullptr; RoleRec* role_rec_o = nullptr;   // [N+1], [R]
    uint32_t* role_cnt_s = nullptr; uint32_t* role_off_s = nullptr;  // [S+1]
    RoleRec*  role_rec_s = nullptr;                                  // [R]
    float4*   role_prm = nullptr;                                    // distinct parameter sets of the bonded terms
    // SPME (mdx_pme.hip)
    float* pme_q = nullptr; float2* pme_f = nullptr; float* pme_theta = nullptr;
    float* pme_q2 = nullptr; float2* pme_f2 = nullptr;   // alchemical window: the coupled molecule's own mesh
    float4* pme_force = nullptr;   // [S] reciprocal-space force when the chain runs on its side stream
    // constraints and virtual sites
    ConsGroup* cons_o = nullptr; ConsGroup* cons_s = nullptr;
    // clusters in slot order (mdx_remap_constraints): staging copy, leaders per tile (a bit per slot), their counts and offsets
    ConsGroup* cons_tmp = nullptr; unsigned long long* cons_mask = nullptr; uint32_t* cons_cnt = nullptr; uint32_t* cons_off = nullptr;
    uint32_t cons_cap_tiles = 0; const uint32_t* cons_n_dev = nullptr;     // cons_n_dev: clusters this handle solves (device word), null = n_groups
    float* cons_vir = nullptr;     // per constraint cluster: r . G of the last SHAKE position stage (kcal/mol); the X-H4 stars behind the others
    ConsStar5* star_o = nullptr; ConsStar5* star_s = nullptr;
    VSite* vsite_o = nullptr; VSite* vsite_s = nullptr;
    uint8_t* wstep_s = nullptr; uint32_t cap_wstep = 0;     // [slots] bit 4: the slot is stepped by water_step_kernel (mixed systems: integrate_kernel skips it)
    GroupSite* gsite_o = nullptr; GroupSite* gsite_s = nullptr; GroupSite* gsite_tmp = nullptr;   // per constraint cluster (null: no cluster carries its site)
    // control / reductions
    StepCtl* ctl = nullptr;
    double*  energy = nullptr;     // [EN_COUNT + 8 + MDX_ESTRIDE*MDX_EPART]: energies, max|F|^2 bits, momentum (px,py,pz,mass),
                                   // then MDX_EPART x {lj, coulomb, virial, -} partial sums of the pair kernel
    uint32_t* flags_dev = nullptr; // misc error flags
    unsigned long long* pair_count = nullptr;  // cluster pairs in the list (statistics)
    unsigned long long* inner_count = nullptr; // [MDX_EPART + 1] dual list: kept cluster pairs per pruning pass (spread), passes
    float*   bbox_red = nullptr;   // [6] min/max reduction (vacuum grid)
    uint32_t* tile_bnd = nullptr;  // [T+1] decomposed handle: 1 <=> the tile holds a ghost or lists a cluster that does
    uint32_t* tile_scan = nullptr; // [T+1]
    uint32_t* tile_order = nullptr;// [T] interior tiles first (launched while the halo message is in flight), then boundary tiles
    uint32_t* tile_lpt = nullptr;  // [T] tiles by decreasing list length (mid-size launches: longest lists first)
    uint32_t cap_tile_lpt = 0;
    float4*  scratch4 = nullptr;   // [cap_scratch4] caller-order scratch (force read-back)
    // fused list rebuild (mdx_grid.hip, rebuild_fast): the tile count and friends stay on the device until the ONE read-back at
    // the end of the chain.  [0] T, [1] first ghost tile (= T without separate ghost columns), [2] non-finite coordinate seen,
    // [3] interior tiles, [4] S + 1 = (T + 1) * 64 + 1 (length of the role-count scan)
    uint32_t* rb_ctl = nullptr;
    unsigned long long* scan_chain = nullptr;   // [64] chained-window scans: (generation << 32) | running total per window, then the windows' done ticks
    // energy between molecules / groups (mdx_groups.hip): group of every GLOBAL atom, and the raw n x n sums
    uint8_t* grp = nullptr; double* grp_mat = nullptr;
    float4* ewald_tab = nullptr;   // Ewald real space: table of the smooth part of the force (mdx_pair_dev.h)
};

struct MdxDecomp;   // mdx_comm.h: the handle is one rank of a spatially decomposed box
struct mdx_handle;
bool mdx_dd_brick_bounds(const mdx_handle* h, int d, float* lo, float* hi);   // this rank's brick in a cut dimension (false: not cut / not decomposed)
bool mdx_dd_half_shell(const mdx_handle* h);   // decomposed with a half-shell halo: every cross-rank pair is evaluated on ONE rank, ghost forces travel back

struct mdx_handle {
    int device = 0;
    MdxDecomp* dd = nullptr;
    hipStream_t stream = nullptr;
    uint32_t N = 0;
    mdx_config cfg{};
    // host copies of the system (needed to rebuild after set_box etc.)
    bool periodic = false;      // any dimension periodic
    int per[3] = {0, 0, 0};     // per-dimension periodicity
    uint32_t n_local = 0, cap_local = 0;  // atoms simulated by this handle (== N on a single GPU)
    uint32_t cap_scratch4 = 0;
    float local_lo[3]{}, local_hi[3]{};   // extent of the local region in non-periodic dimensions (decomposed runs)
    bool have_local_bounds = false;
    float box_lo[3]{}, box_hi[3]{};
    uint32_t n_bonds = 0, n_angles = 0, n_dih = 0, n_p14 = 0;
    uint32_t n_roles = 0;
    uint32_t n_roles_excl = 0;        // ... of which Ewald exclusion corrections (ROLE_EWALD_EXCL)
    uint32_t n_roles_dih = 0;         // ... of which neither bond nor angle roles (none: the fused bonded + kick + drift pass runs a flavour without the other branches)
    bool excl_inside_rigid = false;   // every excluded pair lies inside ONE rigid three-site cluster (its virtual site included)
    uint32_t n_groups = 0, n_cons = 0, n_vsites = 0;   // constraint clusters of up to four atoms / constraints / virtual sites
    uint32_t n_star5 = 0; std::vector<ConsStar5> h_star5;   // X-H4 clusters (caller order)
    bool cons_all_rigid3 = false;    // every constraint cluster is a rigid three-site water: the solvers' register-only flavour
    bool vsites_in_groups = false;   // every virtual site is placed by its parents' constraint cluster (GroupSite)
    uint32_t water_step_launches = 0, water_step_mixed_launches = 0;      // (diagnostics: mdx_pair_launch_info)
    uint32_t n_wstep_groups = 0;          // rigid three-site water clusters the one-pass step kernel may take (ConsGroup::wstep)
    bool wstep_all = false;               // ... and they (with their sites) are every atom of the handle: nothing else to integrate or constrain
    bool wstep_sites_all = false;         // every virtual site belongs to such a cluster: the sites' forces can be left to that kernel
    bool vsite_spread_deferred = false;   // step loop, rigid-water boxes: the force call being enqueued leaves the sites' forces to the next water_step_kernel ...
    bool vsite_spread_pending = false;    // ... and has done so: that kernel spreads them
    bool vsites_fresh = false;       // ... and the last position stage did so: the next force call has nothing to construct
    std::vector<ConsGroup> h_groups; std::vector<VSite> h_vsites;   // host copies (caller order): ownership anchors of a decomposition
    int hc_kind = 1; uint32_t hc_order = 0, hc_iter = 0; std::string hc_text;   // mdx_set_hydrogen_constraint: what the host asked for
    bool cons_dirty = false;                           // positions were set from outside: project them once
    bool vsites_convex = true;                         // every virtual site lies inside the triangle of its parents
    // SPME
    bool pme_on = false; int pme_K[3] = {0, 0, 0}; void* pme_plan = nullptr;  // opaque PmePlan
    // the charge mesh is cleared BEHIND the chain that dirtied it (mdx_pme.hip), not in front of the next one
    bool pme_canvas_clean = false, pme_canvas2_clean = false, pme_clear_pending = false, pme_block_spread_used = false, pme_spread_main = false;
    int pme_cus_per_xcd = 0;     // > 0: the handle's stream is confined to the other 32 - n compute units of every XCD (mdx_pme_cu_split)
    bool pme_overlap = false; hipStream_t stream_pme = nullptr; hipEvent_t ev_pme_fork = nullptr, ev_pme_join = nullptr;
    uint32_t ewald_tab_n = 0, ewald_tab_shift = 16; float ewald_tab_scale = 1.f;
    double ewald_self = 0.0, ewald_background = 0.0; double total_charge = 0.0, sum_q2 = 0.0, q_abs_max = 0.0;
    uint32_t n_mobile = 0;
    double total_mass = 0.0;
    std::vector<uint8_t> flags;
    std::vector<float> h_mass;
    std::vector<float2> h_lj;          // host copy of the per-atom LJ record (sign of .y marks the alchemical molecule)
    std::vector<uint32_t> mol_start;   // first atom of each molecule
    std::vector<uint8_t> grp_host; uint32_t n_grp = 0; bool grp_by_mol = false;   // mdx_set_energy_groups: group of every atom (0: no matrix is kept)
    bool alch_on = false; double alch_lambda = 0.0; uint32_t alch_lo = 0, alch_hi = 0;
    float sc_alpha = 0.5f, sc_sigma_min = 3.0f;   // soft core of the alchemical window (mdx_set_alchemical_softcore)
    // grid
    GridParams grid{};
    uint32_t ncol = 0, ncells = 0;
    uint32_t T = 0;          // real tiles (the null tile is tile T)
    uint32_t S = 0;          // slots = (T+1)*64
    uint32_t cap_tiles = 0;  // allocation capacity in tiles (incl. null)
    size_t cap_scan = 0;
    uint64_t E = 0, cap_entries = 0;
    uint32_t MC = 0, cap_mchunks = 0;
    float r_list = 0.f;
    // dual pair list (rolling pruning inside the pair kernel)
    bool dual_on = false;        // this handle's step loop walks the inner masks
    float inner_skin = 0.f;
    // self-tuning of the library-default buffer (cfg.inner_skin == 0): pruning passes per step over a window of steps
    float inner_skin_auto = 0.f; bool dual_auto_off = false; uint32_t dual_win_steps = 0, dual_win_prunes = 0;
    bool prune_pending = true;   // the next step-loop force call must prune (after a rebuild / at the start of mdx_step)
    uint32_t inner_rebuilds = 0;       // list rebuilds so far whose pruning pass wrote the inner list
    bool inner_from_rebuild = false;   // the last list rebuild produced the inner list itself (prune_list_kernel<true>): the force call behind it walks it
    bool path_split = false;           // the dual list's path accumulators live in d.path / d.dprune (DeviceState)
    OnePassNow onepass;                // one launch per step: the launch mdx_step is enqueuing right now
    bool onepass_refused = false;      // the pair kernel's flavour of this handle has no such instantiation (set by the first attempt)
    uint64_t onepass_launches = 0, onepass_violations = 0;
    bool prune_latch = false;    // ... latched for the (up to two) launches of that force call
    bool moved_outside = true;   // something other than the step loop moved atoms in slot space (minimiser, constraint projection):
                                 // the path accumulators did not see it, the next mdx_step starts with a pruning pass
    // host work to do while the device finishes an energy evaluation (mdx_single_point: the fingerprint of the static arrays);
    // called once, between the read-back's enqueue and the wait for it
    void (*wait_hook)(void*) = nullptr; void* wait_hook_arg = nullptr;
    bool kind_split = false;     // clusters are formed per interaction kind and cluster pairs without a common kind are dropped (mdx_grid.hip)
    uint32_t n_interior = 0;     // decomposed handle: tiles whose lists involve no ghost (0: no split)
    bool tile_split = false;     // tile_order / n_interior describe the current list
    bool tile_lpt_on = false;    // tile_lpt describes the current list
    bool tile_lpt_grouped = false;   // ... ordered by length inside each XCD's contiguous range (no round-robin over the XCDs then)
    bool want_tile_split = false; uint32_t cap_tile_split = 0;   // set by the decomposition (world > 1, overlap on)
    int nb_step = -1;            // chunk step of the force call being enqueued (-1: not from the step loop -> outer masks)
    // Energies at a cadence (mdx_set_energy_cadence, snapshots, the barostat): the force call that ends such a step runs the
    // energy flavour of the kernels (e_pending: its sums sit in d.energy), mdx_finalize_energy_cache adds the kinetic energy and
    // the constraint virial behind the thermostat and keeps the result; mdx_energy at that step returns it.
    uint32_t energy_every = 0; bool e_pending = false, e_cache_valid = false; uint64_t e_cache_step = 0; mdx_energies e_cache{};
    bool nb_post_rebuild = false;   // the force call that finishes a step behind a list rebuild: it is the pruning pass itself
    // step loop: length of the rebuild-free stretches (steps), so that a chunk ends near the step the list is expected to go
    // stale at instead of enqueueing up to chunk_steps - 1 launches the device then gates off
    uint32_t steps_since_rebuild = 0, stretch_samples = 0; float stretch_mean = 0.f, stretch_dev = 0.f;
    // Verlet skin chosen by the library (mdx_config.skin == 0): hill-climb on the measured step rate over windows of several
    // rebuilds (mdx_step).  The skin only decides which pairs are LISTED; forces and trajectories do not depend on it.
    struct SkinTune {
        bool on = false; int phase = 0;          // 0 warm-up, 1 measuring the base, 2 walking down, 3 walking up, 4 done
        float base_skin = 2.f, best_skin = 2.f, trial = 2.f; double base_rate = 0.0, best_rate = 0.0;
        uint32_t win_steps = 0, win_rebuilds = 0, skip_rebuilds = 0, warm_steps = 0;
        float dt_tuned = 0.f;                    // the time step the walk ran (or runs) at
        std::chrono::steady_clock::time_point t0;
    } skin_tune;
    int chunk_s = -1;            // decomposed driver: chunk step whose drift has been enqueued (its prune word is shared by the halo unpack)
    // state flags
    bool list_valid = false;    // spatial caches match the slot-space state
    bool forces_valid = false;
    bool in_slot_space = false; // dynamic state lives in slot arrays (else in *_orig staging)
    bool have_ext = false;
    uint64_t step_count = 0, rebuild_count = 0;
    // profiling
    bool profile = false; int profile_level = 0; bool prof_open = false;
    struct EvPair { hipEvent_t a, b; int kind; int tag; hipStream_t st; };
    int prof_tag = -1;       // chunk step of the launches being enqueued (-1: ungated)
    uint32_t prof_seq = 0; bool prof_sampled = false;     // level 2 brackets every MDX_PROF_SAMPLE-th pair launch (mdx_prof_begin)
    std::vector<EvPair> ev_pending;
    std::vector<hipEvent_t> ev_pool;
    mdx_stats stats{};
    // thermostat / COM / snapshots
    int tstat_kind = 0; float tstat_temp = 300.f, tstat_tau = 1.f; uint32_t tstat_every = 10;
    int integrator = 0; float lang_gamma = 1.f, lang_temp = 300.f; uint64_t lang_seed = 0, lang_step = 0;
    int baro_kind = 0; float baro_p0 = 1.f, baro_tau = 5.f, baro_beta = 4.5e-5f; uint32_t baro_every = 25;
    double last_pressure = 0.0, last_mu = 1.0;
    bool force_zeroed = false;   // the integrate pass just enqueued cleared the force array (half-list kernel: skip the fill)
    bool cons_full_kick = false; // the SHAKE pass about to be enqueued follows a fused full kick (closing + opening): its corrections are those of a force acting through dt
    bool bonded_fused = false;   // the pair launch just enqueued carried the bonded gather in extra workgroups: skip its own launch
    uint32_t pair_info_step[8] = {}, pair_info_any[8] = {};   // the pair-kernel instantiation last launched by the step loop over the dual list / by anything else (mdx_pair_launch_info)
    bool bonded_deferred = false; // step loop, large classes: the NEXT step's fused bonded + kick + drift pass evaluates the bonded terms of this force call
    uint64_t rng_state = 0;
    bool zero_com = false;
    uint32_t snap_every = 0; bool snap_vel = false;
    uint32_t snap_handlers[MDX_SNAP_HANDLERS] = {};   // mdx_set_snapshot_handlers: cadence per handler (0 = off); snap_every stays the plain cadence of mdx_set_snapshot_cadence
    double time_ps = 0.0;
    struct Snapshot { double time; uint64_t step; mdx_energies e; std::vector<float> pos, vel, frc; std::vector<mdx_hbond> hbonds; std::vector<float> between; uint32_t handler_mask = 0; };
    // md.water views and hydrogen-bond detection (mdx_set_water_layout / mdx_set_hbond_detection)
    uint32_t water_first = 0, n_waters = 0, water_sites = 0;
    std::vector<uint8_t> hb_heavy; float hb_dmax = 2.5f, hb_angle_min = 120.f;
    std::vector<uint32_t> hb_donor_of;   // [N] heavy atom a hydrogen is bound to (MDX_INVALID: not a donor hydrogen)
    std::vector<uint32_t> h_bond_pairs;  // bonds + constraints as given at creation (pairs), for the donor table
    std::vector<Snapshot> snapshots;
    DeviceState d;
    StepCtl* h_ctl = nullptr;  // pinned (+ 64 bytes: the sequence word of the chunk-end readback)
    uint32_t ctl_seq = 0;
    uint32_t* h_rb = nullptr;  // pinned, device-visible: the list rebuild's counters land here straight from a kernel
    uint32_t rb_seq = 0;        // sequence number of the last read-back (word 31 of h_rb: the host spins on it)
    uint32_t scan_gen = 0;           // generation stamp of the chained scans (no reset between launches)
    bool cell_count_clean = false;   // fused rebuild: d.cell_count is all zero (the grid-scan kernel zeroes what it has consumed)
    bool slot_of_clean = false;      // ... and d.slot_of holds no slot of an atom that has left the local set
};

// argument block of the fused bonded + kick + drift pass (mdx_integrate.hip)
struct FusedArgs {
    uint32_t S; float dt;
    const float4* posq_in; float4* posq_out; float4* vel; float4* force; float4* ref;
    float* path; const float* dprune;      // path split (DeviceState): null -> the accumulator rides in ref[].w
    const uint32_t* role_off; const RoleRec* roles; const float4* prm; BondedParams p; uint32_t R;
    const uint32_t* gate_in; uint32_t* disp_out; uint32_t thr_bits; uint32_t* prune_out; float path_thr;
    // Decomposed handle, pipelined step (PIPE; mdx_decomp.hip).  The pass takes over what two kernels and two launch boundaries did:
    //   pipe_flags & 1  the ghost forces the peers returned for the previous force call are ADDED here (frc_in has the layout of the
    //                   position message: the rows a slot fills there are the rows that come back for it) - no add kernel, no atomics;
    //                   the peers' "my list went stale" words ride in the flag rows and are merged into the gate
    //   pipe_flags & 2  the halo PACK: a slot's new position goes straight into its rows of the position message; the workgroup that
    //                   finishes last writes the flag rows (the stale word is only complete then) and resets the tile queues of the
    //                   pair launch that follows
    uint32_t pipe_flags;
    const uint32_t* send_cnt; const uint32_t* send_rows;      // per slot: rows of the position message it fills (at most 7: one per peer), [S], [7 S]
    const float4* frc_in; float4* send_buf;
    uint32_t n_flag; uint32_t flag_rows[32];                   // one per peer (DD_MAX_WORLD: mdx_decomp.hip asserts it)
    PipeCtl* pc;
    uint32_t* gate_word;                                       // = gate_in, writable: the merged stale word is stored back for the kernels behind
};
bool mdx_dd_pipe_fill(mdx_handle* h, FusedArgs& a, uint32_t* gate_word);   // decomposed handle in its pipelined arrangement: fills the pipe fields (false: plain pass)
void mdx_dd_note_rebuild(mdx_handle* h);                                     // the slot order changed: per-slot tables of the decomposition are stale

// ---- error plumbing ------------------------------------------------------------------------------
void mdx_set_error(const std::string& s);
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            mdx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));              \
            return (_e == hipErrorOutOfMemory) ? MDX_EOOM : MDX_EDEVICE;                   \
        }                                                                                  \
    } while (0)
#define MDX_TRY(expr) do { int _r = (expr); if (_r != MDX_OK) return _r; } while (0)

// ---- kernel launchers (each .hip file) -------------------------------------------------------------
// grid / list build (mdx_grid.hip)
int mdx_rebuild(mdx_handle* h);
int mdx_unsort_state(mdx_handle* h);  // slot space -> pos_orig / vel_orig
int mdx_gather_to_orig(mdx_handle* h, const float4* slot_arr, float4* orig_arr);
int mdx_extract_neighbors(mdx_handle* h, uint32_t* offsets, uint32_t* idx);
int mdx_classify_tiles(mdx_handle* h, bool by_length);
int mdx_order_tiles_by_length(mdx_handle* h, bool grouped);   // tile_lpt: longest lists first
// forces
// part: 0 = every tile; 1 = the interior tiles of a decomposed handle (no ghost in their lists: they run while the halo
// message is in flight); 2 = its boundary tiles (after the unpack)
int mdx_launch_nonbonded(mdx_handle* h, bool energy, const uint32_t* d_gate, uint32_t thr_bits, int part = 0);
int mdx_build_ewald_table(mdx_handle* h);     // (mdx_nonbonded.hip) at creation: depends on ewald_alpha and the cut-offs
int mdx_launch_bonded(mdx_handle* h, bool energy, const uint32_t* d_gate, uint32_t thr_bits);
bool mdx_bonded_wanted(const mdx_handle* h);                       // there are bonded roles to evaluate in a force call
void mdx_fill_bonded_params(const mdx_handle* h, BondedParams& p);
int mdx_launch_add_ext(mdx_handle* h, const uint32_t* d_gate, uint32_t thr_bits);
// integration (mode: 0 = half kick + drift, 1 = full kick + drift (also: one leapfrog step), 2 = closing half
// kick, 3 = one Langevin-middle step: full kick, half drift, friction + noise, half drift)
int mdx_launch_integrate(mdx_handle* h, int mode, float dt, const uint32_t* d_gate_in,
                         uint32_t* d_disp_out, uint32_t thr_bits, uint32_t* d_prune_out = nullptr, bool skip_wstep = false);
// one pass for the bonded gather of step s-1's positions, the full kick and the drift of step s (velocity Verlet inside a
// chunk, no constraints / virtual sites / SPME / external forces, one lane per atom): reads posq, writes posq_alt, swaps
bool mdx_bonded_integrate_ok(const mdx_handle* h);
int mdx_launch_bonded_integrate(mdx_handle* h, float dt, const uint32_t* d_gate_in, uint32_t* d_disp_out, uint32_t thr_bits,
                                uint32_t* d_prune_out);
// One launch per step (round 6): the pair launch of the one-wave-per-tile class also finishes the previous step for its tile's atoms and
// evaluates their bonded roles (mdx_nonbonded_impl.h STEP); a chunk opens with mdx_launch_step_begin and, where a launch was gated off,
// returns to the plain form through mdx_launch_step_materialise.  mdx_step (mdx_api.hip) owns the buffer rotation.
bool mdx_onepass_ok(const mdx_handle* h);
int mdx_launch_step_begin(mdx_handle* h, float dt, const uint32_t* d_gate_in, uint32_t* d_disp_out, uint32_t thr_bits, uint32_t* d_prune_out);
int mdx_launch_step_materialise(mdx_handle* h, float dt, const float4* y, const float4* fprev, float4* x_out, bool kick_done, float kick,
                                const uint32_t* d_gate_in, uint32_t thr_bits);
int mdx_launch_kinetic(mdx_handle* h);  // energy[EN_KIN], energy[EN_COUNT] = max |F|^2
int mdx_exclusive_scan_u32(mdx_handle* h, const uint32_t* in, uint32_t* out, uint32_t n);
int mdx_exclusive_scan_u32_ex(mdx_handle* h, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums);   // caller's scratch: n / 2048 + 1 words

// pair-kernel variant actually used: 1 = whole-tile kernel (full list), 2 = cluster-masked kernel (full
// list, deterministic), 3/4 = the same with 1/4 waves per tile forced, 5 = cluster-masked kernel over a
// HALF list: each cluster pair is evaluated once and the reaction force goes back to the j-atoms with
// f32 atomics (half the arithmetic; summation order, hence the last bits, vary run to run).
#define MDX_NB_DEFAULT_VARIANT 5
static inline int mdx_nb_variant(const mdx_handle* h) {
    const uint32_t v = h->cfg.nb_variant;
    if (h->alch_on) return 5;   // the alchemical flavour exists for the half-list kernel only
    return (v >= 1 && v <= 5) ? (int)v : MDX_NB_DEFAULT_VARIANT;
}
static inline bool mdx_nb_half(const mdx_handle* h) { return mdx_nb_variant(h) == 5; }
// tiles below which the half-list pair kernel runs eight waves per tile (and carries the bonded gather in its launch)
// (round 3: a decomposed handle used to keep eight up to 4096 tiles; with its tiles launched longest list first, four waves
// per tile and a bonded launch of its own are faster there too: rank 0 of 8 of the 1 M-atom box 0.108 -> 0.089 + 0.009 ms)
static inline uint32_t mdx_wpt8_below(const mdx_handle* h) {
    static const int env = [] { const char* e = std::getenv("MDX_WPT8_BELOW"); return e ? std::atoi(e) : -1; }();
    (void)h;
    return env >= 0 ? (uint32_t)env : 2048u;
}

// Least number of waves in a workgroup of the cluster pair kernel: with w waves per tile a workgroup holds max(w, this) waves,
// i.e. max(1, this / w) tiles.  It was 4 (a workgroup of 256 threads: two or four tiles at one or two waves per tile); measured at
// the end of round 3 with one wave per tile at water1M: 4 -> 0.4600 ms per pair launch, 2 -> 0.4560, 8 -> 0.4753, 1 -> 0.4487 - a
// one-wave workgroup has no barrier, frees its slot the moment its tile is done and is dispatched one tile at a time.
#ifndef MDX_NB_WAVES
#define MDX_NB_WAVES 1
#endif
// waves per tile of the half-list pair kernel for a list of T tiles (MDX_WPT: A/B knob).  The rule is evaluated on the host
// (launch geometry) AND on the device (the fused list rebuild orders the tiles by length inside the pair kernel's XCD ranges
// before the host knows T), so it is one __host__ __device__ function of a small parameter block.
struct WptRule { int env; uint32_t wpt8_below; int decomposed; };
__host__ __device__ static inline int mdx_wpt_rule(const WptRule& r, uint32_t T) {
    if (r.env) return r.env;
    // With the tiles launched longest list first (round 3) fewer waves per tile win further down than before: one wave from
    // 12 k tiles on (water1M: pair launch 0.464 -> 0.458 ms, 1844 -> 1870 steps/s; it was two), two from 3 k tiles on for a
    // single-device list (375 k atoms 0.213 -> 0.197 ms, 585 k atoms 0.312 -> 0.289; it was four up to 12 k tiles)
    // (a decomposed rank - owned bricks and halo shells, very uneven lists - keeps four up to 8 k tiles: rank 0 of 4, 5.5 k tiles,
    // 0.155 ms with four against 0.180 with two; rank 0 of 2, 9.4 k tiles, 0.285 with four against 0.257 with two)
    if (T >= 12000u) return r.decomposed ? 2 : 1;
    if (T >= (r.decomposed ? 8000u : 3000u)) return 2;
    return T < r.wpt8_below ? 8 : 4;
}
static inline WptRule mdx_wpt_rule_of(const mdx_handle* h) {
    static const int env = [] { const char* e = std::getenv("MDX_WPT"); return (e && (e[0] == '1' || e[0] == '2' || e[0] == '4' || e[0] == '8')) ? e[0] - '0' : 0; }();
    WptRule r; r.env = env; r.wpt8_below = mdx_wpt8_below(h); r.decomposed = (h->dd || h->n_local != h->N) ? 1 : 0;
    return r;
}
static inline int mdx_nb_wpt_half(const mdx_handle* h, uint32_t T) { return mdx_wpt_rule(mdx_wpt_rule_of(h), T); }

// constraints / virtual sites (mdx_constraints.hip)
int mdx_build_constraints(mdx_handle* h, const mdx_system* s);
int mdx_remap_constraints(mdx_handle* h);                  // caller order -> slot order, at every rebuild
int mdx_launch_constrain_positions(mdx_handle* h, float dt, const uint32_t* d_gate, uint32_t* d_disp_out, uint32_t thr,
                                   uint32_t* d_prune_out = nullptr, bool skip_wstep = false);
int mdx_launch_constrain_velocities(mdx_handle* h, const uint32_t* d_gate, uint32_t thr);
bool mdx_water_step_ok(const mdx_handle* h);
bool mdx_water_step_mixed(const mdx_handle* h);          // ... beside other mobile atoms / clusters: they keep integrate_kernel and the cluster solvers               // a box of rigid water: spread + kick + drift + SETTLE + site placement as ONE pass per step
int mdx_launch_water_step(mdx_handle* h, int mode, float dt, const uint32_t* d_gate, uint32_t* d_disp_out, uint32_t thr, uint32_t* d_prune_out);
int mdx_launch_vsite_construct(mdx_handle* h, const uint32_t* d_gate, uint32_t thr);
int mdx_launch_constraint_virial(mdx_handle* h);
int mdx_check_box(const mdx_handle* h, const float* lo, const float* hi);   // the checks mdx_set_box applies   // energy[EN_VIRIAL] += sum of cons_vir
int mdx_launch_vsite_spread(mdx_handle* h, const uint32_t* d_gate, uint32_t thr);

// SPME reciprocal space (mdx_pme.hip)
int mdx_pme_setup(mdx_handle* h);
int mdx_pme_cu_split(const mdx_handle* h, const mdx_config* c);      // compute units per XCD the reciprocal-space chain gets for itself (0: none)
int mdx_stream_create_masked(hipStream_t* s, int cus_per_xcd, bool complement);
int mdx_stream_unmask(mdx_handle* h);                                // the handle's stream back on every compute unit (idle handle)                      // plans, mesh, theta table (again after set_box)
void mdx_pme_destroy(mdx_handle* h);
int mdx_launch_pme(mdx_handle* h, bool energy, const uint32_t* d_gate, uint32_t thr);
int mdx_pme_fork(mdx_handle* h);                                             // side stream starts behind the handle's stream
int mdx_pme_join(mdx_handle* h, const uint32_t* d_gate, uint32_t thr);       // ... and hands its force back

// shared between mdx_api.hip and mdx_extras.hip
int mdx_compute_forces(mdx_handle* h, bool energy, const uint32_t* gate, uint32_t thr);
int mdx_ensure_ready(mdx_handle* h);
int mdx_energy_impl(mdx_handle* h, mdx_energies* out);
int mdx_after_steps(mdx_handle* h, float dt, uint32_t done);      // thermostat / COM / snapshots at their cadence
uint32_t mdx_steps_to_next_event(const mdx_handle* h);            // chunk lengths stop at these boundaries
bool mdx_energy_wanted_at(const mdx_handle* h, uint64_t step);    // does something read the energies after step `step`?
int mdx_finalize_energy_cache(mdx_handle* h);                     // e_pending -> e_cache (kinetic energy, constraint virial, read-back)
int mdx_groups_evaluate(mdx_handle* h, float* out /* [n_grp^2] */);   // energy_potential_between_mols of the current state (mdx_groups.hip)
int mdx_launch_scale_velocities(mdx_handle* h, float lambda, const double* com_v_or_null);
int mdx_launch_momentum(mdx_handle* h);                           // energy[EN_COUNT+1..] <- sum m v (3 doubles) + mass

static inline bool mdx_has_constraints(const mdx_handle* h) { return h->n_groups != 0 || h->n_star5 != 0; }
// degrees of freedom: 3 per mobile atom, minus constraints, minus the centre of mass
static inline double mdx_dof(const mdx_handle* h) {
    const double d = 3.0 * (double)h->n_mobile - (double)h->n_cons - 3.0;
    return d < 1.0 ? 1.0 : d;
}

// counter-based RNG shared (bit for bit) with the oracle
__host__ __device__ static inline uint64_t mdx_splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// rocprof markers: roctxRangePush / roctxRangePop around the phases of the step loop, so that
// `rocprofv3 --marker-trace` shows rebuild / halo / force phases beside the kernels.  librocprofiler-sdk-roctx (or the older
// libroctx64) is dlopen'd on first use, only when MDX_ROCTX=1 is set; otherwise - and when neither library is there - the
// calls are no-ops (one predictable branch).
void mdx_range_push(const char* name);
void mdx_range_pop();
struct MdxRange {
    explicit MdxRange(const char* name) { mdx_range_push(name); }
    ~MdxRange() { mdx_range_pop(); }
    MdxRange(const MdxRange&) = delete; MdxRange& operator=(const MdxRange&) = delete;
};

// profiling helpers
// kinds: 0 pair kernel (whole / interior half), 1 bonded, 2 integrate, 3 energy flavour, 4 pair kernel boundary half, 5 fused bonded +
// kick + drift; profile level 3 (a decomposed step in its production arrangement, phase by phase) adds 6 halo pack, 7 halo wire,
// 8 halo unpack, 9 ghost-force pack, 10 ghost-force wire, 11 ghost-force add.  st: the stream the work goes to (default: h->stream)
void mdx_prof_begin(mdx_handle* h, int kind, hipStream_t st = nullptr);
void mdx_prof_end(mdx_handle* h);
void mdx_prof_collect(mdx_handle* h, int first_stale_step = 1 << 30);
