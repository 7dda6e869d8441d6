// mdx_comm.h — multi-GPU plumbing below the C ABI (not part of the public ABI): the transport a decomposed handle
// talks through and the per-handle decomposition state.  SURVEY.md §8e: spatial decomposition, ghost-atom halo
// exchange with ncclSend / ncclRecv groups on a dedicated communication stream, interior tiles computed while the
// halo is in flight.  The reference is single-device (/root/reference src/util.rs:1086), so this is new capability.
#pragma once
#include "mdx_internal.h"
#include <vector>

struct MdxSeg { int peer; uint32_t row0, nrows; };   // rows are float4

// What a decomposed handle needs from the wire.  Two implementations (mdx_comm.hip): RCCL over xGMI (one process or
// thread per GPU, librccl dlopen'd on first use) and an in-process fabric (N handles of ONE process, any mix of
// devices: plain device-to-device copies between their buffers; what the single-GPU tests and a single-process host use).
struct MdxTransport {
    int rank = 0, world = 1;
    virtual ~MdxTransport() {}
    // Enqueue on `stream`: rows [row0, row0 + nrows) of `send` travel to `peer` for every send segment, the rows of
    // every receive segment land in `recv`.  A peer's send segment towards this rank and this rank's receive segment
    // from that peer have the same length by construction.  A segment whose peer is this rank is a local copy.
    virtual int exchange(const float4* send, const std::vector<MdxSeg>& ssegs, float4* recv, const std::vector<MdxSeg>& rsegs,
                         hipStream_t stream) = 0;
    // In-place all-reduce of a small DEVICE array (n <= 4096).  kind: 0 = sum of doubles, 1 = max of uint32.
    virtual int all_reduce(void* dev, size_t n, int kind, hipStream_t stream) = 0;
    // In-place sum of a LARGE device array of floats over the ranks (the SPME charge mesh of a decomposed handle).
    virtual int all_reduce_f32(float* dev, size_t n, hipStream_t stream) = 0;
    // every rank contributes one word, every rank gets all of them (host values; synchronises `stream`)
    virtual int all_gather_u32(uint32_t mine, uint32_t* all, hipStream_t stream) = 0;
    virtual const char* name() const = 0;
    virtual bool delivers() const { return true; }   // false: the null transport (peers' rows never arrive)
    // the null transport with MDX_NULL_WIRE_US set: every send/recv group is a device-side wait of that many microseconds on the stream
    // it is enqueued on, and the handle keeps its receive buffers filled with what the unpack / add kernels turn into no-ops - one rank of N
    // measured alone then runs every kernel of the real step and pays a stated wire time (tools/one_rank_profile.py)
    virtual bool loopback() const { return false; }
    // Does the handle choose between the half-shell halo (two messages per step) and the full shell (one) from this transport's
    // measured message time (mdx_dd_attach)?  The wire (RCCL) and the null transport with a stated wire time do; the in-process
    // fabric and the shared-memory staging - verification aids, their "wire" is host synchronisation - keep the half shell.
    virtual bool wire_time_decides() const { return false; }
    // RCCL only: ncclGetVersion and ncclCommCount of this handle's communicator (0, 0 elsewhere) - for mdx_comm_diag
    virtual void wire_info(int* version, int* comm_count) const { *version = 0; *comm_count = 0; }
};

struct mdx_fabric;   // the in-process transport's meeting point (C ABI: mdx_fabric_create / _destroy)
MdxTransport* mdx_make_rccl_transport(const uint8_t* id128, int rank, int world, int device);    // nullptr + error text on failure
MdxTransport* mdx_make_fabric_transport(mdx_fabric* f, int rank);
MdxTransport* mdx_make_shm_transport(const char* name, int rank, int world);   // processes of one host, rows staged through POSIX shared memory
MdxTransport* mdx_make_null_transport(int rank, int world);   // delivers nothing: one rank of N profiled alone (tools/one_rank_profile.py)

// ---- decomposition state of one handle -------------------------------------------------------------------------
struct MdxDecomp {
    MdxTransport* tr = nullptr;
    int rank = 0, world = 1;
    int grid[3] = {1, 1, 1}, coord[3] = {0, 0, 0};
    float box_lo[3]{}, box_len[3]{};
    float brick_lo[3]{}, brick_hi[3]{};
    float r_list = 0.f, margin = 0.f, ext = 0.f, halo = 0.f;   // halo = r_list + margin + ext
    // replicated static data
    uint32_t* anchor = nullptr;        // [N] the atom whose position decides the owner of atom g (constraint cluster / virtual-site parent)
    // global dynamic state at the last gather (device, replicated)
    float4* g_pos = nullptr; float4* g_vel = nullptr; float4* g_frc = nullptr;   // [N]
    // classification scratch
    uint8_t* cls = nullptr;            // [N] 0: not here, 1: owned, 2: ghost
    uint8_t* owner = nullptr;          // [N] owning rank
    uint8_t* shift_code = nullptr;     // [N] image that brings the atom into this rank's frame: (kx+1) | (ky+1) << 2 | (kz+1) << 4
    uint32_t* send_mask = nullptr;     // [N] owned atoms: bit q <=> rank q keeps a ghost copy
    uint32_t* flags = nullptr; uint32_t* scan = nullptr; uint32_t* scan_sums = nullptr; size_t cap_flags = 0;
    // local atom set (ascending global id)
    uint32_t n_local = 0, n_owned = 0, cap_local = 0;
    uint32_t* gid_local = nullptr; uint8_t* ghost_local = nullptr; float4* pos_l = nullptr; float4* vel_l = nullptr;
    float4* pos_at_part = nullptr;     // local positions at the last repartition ("is the local set still complete?")
    uint32_t* owned_gid = nullptr;     // [n_owned]
    // halo lists: all peers' rows in ONE send and ONE receive buffer, a peer's segment ends with a flag row (id 0xFFFFFFFF)
    uint32_t* send_ids = nullptr; uint32_t* recv_ids = nullptr; float4* recv_shift = nullptr;
    float4* send_buf = nullptr; float4* recv_buf = nullptr;
    uint32_t n_send = 0, n_recv = 0, cap_send = 0, cap_recv = 0;
    std::vector<MdxSeg> send_segs, recv_segs;
    // Half-shell halo (default with the half-list pair kernel): a rank keeps ghosts only of atoms whose OWNER's brick lies in
    // an upper direction (first non-zero component of the brick offset positive), so a pair of atoms owned by two ranks is
    // evaluated on exactly one of them, and the forces that rank computed on its ghosts travel back along the same
    // segments in the opposite direction: frc_send has the layout of recv_buf, frc_recv that of send_buf.
    bool half_shell = false;
    float wire_us = -1.f;              // one send/recv group of a typical halo message on this transport, measured at attach (max over the ranks); < 0: not measured
    float4* frc_send = nullptr; float4* frc_recv = nullptr;
    bool force_return_pending = false;   // begin() was enqueued, end() has not been yet
    // gather of the global state (repartition, read-back)
    float4* gat_send = nullptr; float4* gat_recv = nullptr; size_t cap_gat_send = 0, cap_gat_recv = 0;
    // communication stream and the events that order it with the compute stream
    hipStream_t comm_stream = nullptr; hipEvent_t ev_packed = nullptr, ev_arrived = nullptr;
    // the interior tiles' pair kernel runs on a side stream, beside the unpack + boundary tiles of the main stream (two
    // half-size launches back to back would each pay their own tail)
    hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_interior = nullptr;
    bool halo_pending = false;         // the next force call of the step loop starts with a halo exchange
    int halo_step = -1;                // chunk step of that exchange (flag word = step + 1)
    bool overlap = true;               // interior tiles run (on the side stream) while the message is packed, sent and unpacked
    // ... or not: whether the split pays depends on the wire time it hides against the fork / join it costs, so a handle
    // tries both over its first chunks and keeps the faster (a local scheduling choice: ranks need not agree).
    // MDX_HALO_OVERLAP=0 / 1 pins it.
    int tune_phase = 0;                // 0: measuring with the split, 1: without, 2: decided
    double tune_ms[2] = {0.0, 0.0}; uint32_t tune_steps[2] = {0, 0}; uint32_t tune_chunks = 0;
    double* red = nullptr;             // [64] device scratch of the small all-reduces
    // chunk end (step loop): the largest displacement since the last repartition is measured and all-reduced BEHIND the chunk's
    // last kernel, every chunk, and comes back with the step-control words - a stale list then needs no second round trip to
    // decide between a local rebuild and a repartition (it was: drift kernel, all-reduce, copy, synchronise - ~80 us per rebuild)
    uint32_t* drift_bits = nullptr;    // [2] device
    bool spec_valid = false; uint32_t spec_bits = 0; uint64_t spec_checks = 0;
    bool probe_both_buffers = false;   // the chunk just enqueued ran fused bonded + kick + drift passes: which of the two position buffers holds the
                                       // state of the step the list went stale at depends on the parity of the gated-off passes behind it
    // ---- the fused bonded + kick + drift pass carries halo pack and ghost-force add (mdx_decomp.hip "fold") ------------------------------
    bool fold_ok = false;              // the drift pass may pack the halo and add the returned ghost forces (half shell, communication on the compute stream, MDX_HALO_FOLD != 0)
    bool rows_fit = true;              // this partition sends no atom to more than seven peers (dd_classify_kernel): the per-slot row table of the fold holds them all
    bool pipe_now = false;             // the step being enqueued has its halo packed by the drift pass (mdx_step)
    bool packed_by_drift = false;      // ... and that pass has been enqueued: the halo exchange starts at the send/recv group
    bool frc_deferred = false;         // the ghost forces returned for the last force call are still in frc_recv: the next drift pass adds them
    uint32_t pipe_gen = 0;             // generation of the last pipelined step enqueued
    PipeCtl* pipe_ctl = nullptr;
    uint32_t* send_cnt = nullptr; uint32_t* send_rows = nullptr; uint32_t cap_rows_slots = 0; bool rows_valid = false;   // per slot: its rows of the position message
    uint64_t pipe_steps = 0;
    // statistics
    uint64_t repartitions = 0, local_rebuilds = 0; uint32_t local_rebuilds_since = 0;
    double repartition_ms = 0.0;
    // mdx_profile(h, 3): GPU time of the phases of a decomposed step, in the production arrangement (mdx_comm_diag)
    double phase_ms[MDX_DIAG_PHASES] = {}; uint64_t phase_n[MDX_DIAG_PHASES] = {};
};

// Is the pair kernel of this step-loop force call launched as interior + boundary halves?  ONE predicate for everybody who
// must agree on it: the launch itself, and the halo unpack (which then raises the ghosts' own prune word).
static inline bool mdx_dd_split_now(const mdx_handle* h) {
    return h->dd && h->tile_split && h->dd->overlap && h->dd->world > 1 && mdx_nb_variant(h) >= 2 && !h->pme_on && (!h->profile || h->profile_level == 3);
}
// Does the fused bonded + kick + drift pass of a decomposed handle carry the halo pack and the add of the returned ghost forces?
// (two kernels and two launch boundaries less per step, whichever way the pair kernel is launched)
static inline bool mdx_dd_fold_eligible(const mdx_handle* h) {
    return h->dd && h->dd->fold_ok && h->dd->rows_fit && h->dd->world > 1 && mdx_nb_half(h) && !h->pme_on && !h->alch_on;
}

int  mdx_set_local_atoms_impl(mdx_handle* h, uint32_t n_local, const uint32_t* d_gid, const uint8_t* d_ghost, const float* d_pos4,
                              const float* d_vel4, const float lo[3], const float hi[3], int32_t periodic);
int  mdx_dd_attach(mdx_handle* h, MdxTransport* tr);     // takes ownership of `tr`; partitions and builds the local state
void mdx_dd_destroy(mdx_handle* h);
int  mdx_dd_halo_begin(mdx_handle* h);                   // pack + exchange (async on the comm stream)
int  mdx_dd_halo_end(mdx_handle* h);                     // wait + unpack: ghost positions, peers' flag words
void mdx_dd_pipe_chunk_end(mdx_handle* h);              // a chunk has been enqueued: nothing is deferred across it
int  mdx_dd_force_return_begin(mdx_handle* h, int flag_word);   // half shell: pack the ghosts' forces + exchange (reverse of the halo)
int  mdx_dd_force_return_end(mdx_handle* h, int flag_word);     // ... and add what came back to the owned atoms
int  mdx_dd_chunk_end_probe(mdx_handle* h, const uint32_t** word_out);   // enqueue the speculative drift probe; *word_out: the device word to read back
int  mdx_dd_on_stale(mdx_handle* h);                     // the list went stale somewhere: local rebuild or repartition (same branch on every rank)
int  mdx_dd_allreduce_host(mdx_handle* h, double* v, int n, bool max_u32 = false);
int  mdx_dd_allreduce_dev(mdx_handle* h, double* dev, size_t n);   // in-place sum of a device array of doubles over the ranks (any n; produced on the handle's stream)
int  mdx_dd_allreduce_f32(mdx_handle* h, float* dev, size_t n, hipStream_t produced_on);   // sum of a large device array over the ranks
int  mdx_dd_exchange(mdx_handle* h, const float4* send, const std::vector<MdxSeg>& ssegs, float4* recv, const std::vector<MdxSeg>& rsegs,
                     hipStream_t produced_on);                // one send/recv group, ordered against `produced_on`
int  mdx_dd_download(mdx_handle* h, int which, float* dst);   // collective: the global array on every rank
int  mdx_dd_gather_global(mdx_handle* h, bool with_force);    // g_pos / g_vel (/ g_frc) <- all ranks' owned atoms
int  mdx_dd_rescale_box(mdx_handle* h, const float hi[3], float mu);   // barostat: scale the gathered state about box_lo, new box, repartition
// host mutation of a joined handle (collective: every rank passes the same data) - md.atoms[i].posit = ..; md.rebuild_spatial_caches()
// (/root/reference src/properties/sol_shrinking_box.rs:599-632), the docking pose loop (src/docking/mod.rs:235)
int  mdx_dd_upload(mdx_handle* h, int which, uint32_t first, uint32_t count, const float4* host_rows);
int  mdx_dd_set_box(mdx_handle* h, const float lo[3], const float hi[3], const float* centre_or_null, const float* mu_or_null);
int  mdx_dd_save_global(mdx_handle* h, float4* backup);      // minimiser: the accepted state (gathered global positions) ...
int  mdx_dd_restore_global(mdx_handle* h, const float4* backup);   // ... and back to it (repartition from the copy)
