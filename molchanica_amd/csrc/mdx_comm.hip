// mdx_comm.hip — the wire under a decomposed handle.
//
//   RcclTransport    one rank per GPU (process or thread).  librccl is dlopen'd on first use, the way hipFFT is
//                    (libmdx.so links no communication library); a halo exchange is ONE ncclGroupStart / ncclSend +
//                    ncclRecv per peer / ncclGroupEnd on the stream it is given (the handle's communication stream).
//                    xGMI is point to point - 7 links per MI355X - and a 2x2x2 decomposition has exactly 7 peers per
//                    rank, so every message has a link of its own; nothing here is a ring collective except the
//                    few-word all-reduces (energies, "is the local atom set still complete").
//   FabricTransport  N handles inside ONE process (threads), on one device or several: a rank posts its send buffer,
//                    peers copy their segments out of it device-to-device.  Same interface, same call sequence, no
//                    RCCL: what the single-GPU tests drive (RCCL refuses two ranks per device) and what a
//                    single-process host with several devices can use.
//   NullTransport    delivers nothing (one rank of N measured alone).
// The reference has no multi-device code at all (/root/reference src/util.rs:1086: CudaContext::new(0)).
#include "mdx_comm.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <cstring>
#include <mutex>

#define FAIL(code, msg) do { mdx_set_error(msg); return (code); } while (0)

// ---- RCCL through dlopen ----------------------------------------------------------------------------------------
namespace {
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclMax = 2 };
enum { ncclUint32 = 3, ncclFloat32 = 7, ncclFloat64 = 8 };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*CommCount)(const ncclComm_t, int*) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

bool load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { mdx_set_error(std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?")); return false; }
    RcclApi a; a.lib = lib;
#define SYM(field, name) *(void**)(&a.field) = dlsym(lib, name); if (!a.field) { mdx_set_error("librccl lacks " name); return false; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    *(void**)(&a.GetVersion) = dlsym(lib, "ncclGetVersion");      // (diagnostics only: absent symbols read as 0)
    *(void**)(&a.CommCount) = dlsym(lib, "ncclCommCount");
    g_rccl = a;
    return true;
}

#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        const int _r = (expr);                                                                                \
        if (_r != ncclSuccess) {                                                                              \
            mdx_set_error(std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error")); \
            return MDX_EDEVICE;                                                                               \
        }                                                                                                     \
    } while (0)

struct RcclTransport : MdxTransport {
    ncclComm_t comm = nullptr;
    uint32_t* d_words = nullptr;   // [world + 1] all-gather scratch
    bool dead = false;             // a call inside a send/recv group failed: the communicator is not used (or destroyed) again
    ~RcclTransport() override {
        // a communicator whose group failed may have operations queued that never complete: ncclCommDestroy would wait
        // for them, so it is abandoned instead (the process is about to report the error and fall back or exit)
        if (comm && !dead) (void)g_rccl.CommDestroy(comm);
        if (d_words) (void)hipFree(d_words);
    }
    const char* name() const override { return "rccl"; }
    bool wire_time_decides() const override { return true; }
    void wire_info(int* version, int* comm_count) const override {
        *version = 0; *comm_count = 0;
        if (g_rccl.GetVersion) (void)g_rccl.GetVersion(version);
        if (g_rccl.CommCount && comm && !dead) (void)g_rccl.CommCount(comm, comm_count);
    }
    int exchange(const float4* send, const std::vector<MdxSeg>& ssegs, float4* recv, const std::vector<MdxSeg>& rsegs,
                 hipStream_t stream) override {
        if (dead) FAIL(MDX_EDEVICE, "rccl transport: an earlier send/recv group failed");
        NCCL_TRY(g_rccl.GroupStart());
        // Inside the group nothing may return early: a thread that leaves with its group open queues every later RCCL
        // call - including ncclCommDestroy - into a group that is never closed.  The first error is remembered, the
        // group is always closed, and the transport is marked dead.
        int first = ncclSuccess, bad_peer = -1; uint32_t bad_rows = 0;
        const char* what = "";
        for (const MdxSeg& s : ssegs) {
            if (!s.nrows || first != ncclSuccess) continue;
            first = g_rccl.Send(send + s.row0, (size_t)s.nrows * 4, ncclFloat32, s.peer, comm, stream);
            if (first != ncclSuccess) { what = "ncclSend"; bad_peer = s.peer; bad_rows = s.nrows; }
        }
        for (const MdxSeg& r : rsegs) {
            if (!r.nrows || first != ncclSuccess) continue;
            first = g_rccl.Recv(recv + r.row0, (size_t)r.nrows * 4, ncclFloat32, r.peer, comm, stream);
            if (first != ncclSuccess) { what = "ncclRecv"; bad_peer = r.peer; bad_rows = r.nrows; }
        }
        const int end = g_rccl.GroupEnd();
        if (first == ncclSuccess && end != ncclSuccess) { first = end; what = "ncclGroupEnd"; }
        if (first != ncclSuccess) {
            dead = true;
            // rank, peer and call: the first real multi-rank run happens on a box nobody watches
            char where[160];
            std::snprintf(where, sizeof(where), "rank %d of %d: %s%s%s (send/recv group of %zu + %zu segments", rank, world, what,
                          bad_peer >= 0 ? " peer " : "", bad_peer >= 0 ? std::to_string(bad_peer).c_str() : "", ssegs.size(), rsegs.size());
            mdx_set_error(std::string(where) + (bad_peer >= 0 ? ", " + std::to_string(bad_rows) + " float4 rows" : std::string()) + "): " +
                          (g_rccl.GetErrorString ? g_rccl.GetErrorString(first) : "rccl error"));
            return MDX_EDEVICE;
        }
        return MDX_OK;
    }
    int all_reduce(void* dev, size_t n, int kind, hipStream_t stream) override {
        if (dead) FAIL(MDX_EDEVICE, "rccl transport: an earlier send/recv group failed");
        NCCL_TRY(g_rccl.AllReduce(dev, dev, n, kind == 0 ? ncclFloat64 : ncclUint32, kind == 0 ? ncclSum : ncclMax, comm, stream));
        return MDX_OK;
    }
    int all_reduce_f32(float* dev, size_t n, hipStream_t stream) override {
        if (dead) FAIL(MDX_EDEVICE, "rccl transport: an earlier send/recv group failed");
        NCCL_TRY(g_rccl.AllReduce(dev, dev, n, ncclFloat32, ncclSum, comm, stream));
        return MDX_OK;
    }
    int all_gather_u32(uint32_t mine, uint32_t* all, hipStream_t stream) override {
        if (dead) FAIL(MDX_EDEVICE, "rccl transport: an earlier send/recv group failed");
        HIP_TRY(hipMemcpyAsync(d_words + world, &mine, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        NCCL_TRY(g_rccl.AllGather(d_words + world, d_words, 1, ncclUint32, comm, stream));
        HIP_TRY(hipMemcpyAsync(all, d_words, sizeof(uint32_t) * world, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return MDX_OK;
    }
};
}  // namespace

MdxTransport* mdx_make_rccl_transport(const uint8_t* id128, int rank, int world, int device) {
    if (!load_rccl()) return nullptr;
    if (hipSetDevice(device) != hipSuccess) { mdx_set_error("hipSetDevice failed"); return nullptr; }
    RcclTransport* t = new RcclTransport();
    t->rank = rank; t->world = world;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, 128);
    const int r = g_rccl.CommInitRank(&t->comm, world, id, rank);
    if (r != ncclSuccess) {
        mdx_set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
        t->comm = nullptr; delete t; return nullptr;
    }
    if (hipMalloc((void**)&t->d_words, sizeof(uint32_t) * (world + 1)) != hipSuccess) { mdx_set_error("hipMalloc failed"); delete t; return nullptr; }
    return t;
}

extern "C" int mdx_comm_unique_id(uint8_t id[128]) {
    if (!id) FAIL(MDX_EPARAM, "null argument");
    if (!load_rccl()) return MDX_EDEVICE;
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, 128);
    return MDX_OK;
}

// ---- in-process fabric ------------------------------------------------------------------------------------------
struct mdx_fabric {
    int world = 0;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; uint64_t generation = 0; bool aborted = false;
    struct Post { const float4* send = nullptr; const std::vector<MdxSeg>* ssegs = nullptr; const void* host = nullptr; uint32_t word = 0; };
    std::vector<Post> post;
    int taken = 0;
    // returns false when the fabric was aborted (a rank failed: nobody may wait for it for ever)
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || aborted; });
        return !aborted;
    }
    void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }
};

extern "C" mdx_fabric* mdx_fabric_create(int world) {
    if (world < 1 || world > 64) { mdx_set_error("fabric world must be in 1..64"); return nullptr; }
    mdx_fabric* f = new (std::nothrow) mdx_fabric();
    if (!f) return nullptr;
    f->world = world; f->post.resize(world);
    return f;
}
extern "C" void mdx_fabric_destroy(mdx_fabric* f) { delete f; }
extern "C" void mdx_fabric_abort(mdx_fabric* f) { if (f) f->abort(); }

namespace {
struct SumPtrs { const float* p[32]; int n; };
__global__ __launch_bounds__(256) void fabric_sum_kernel(size_t n, SumPtrs src, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int q = 0; q < src.n; ++q) s += src.p[q][i];     // rank order: every rank gets the same bits
        out[i] = s;
    }
}

struct FabricTransport : MdxTransport {
    mdx_fabric* f = nullptr;
    float* tmp = nullptr; size_t cap_tmp = 0;
    bool dead = false;      // an exchange named a peer outside the communicator: refused before the barrier, nothing is queued again
    ~FabricTransport() override { if (tmp) (void)hipFree(tmp); }
    int all_reduce_f32(float* dev, size_t n, hipStream_t stream) override {
        if (n > cap_tmp) {
            if (tmp) (void)hipFree(tmp);
            tmp = nullptr; cap_tmp = 0;
            if (hipMalloc((void**)&tmp, sizeof(float) * n) != hipSuccess) return fail();
            cap_tmp = n;
        }
        if (hipStreamSynchronize(stream) != hipSuccess) return fail();
        f->post[rank].host = dev;
        if (!f->barrier()) return fail();
        SumPtrs sp{}; sp.n = world;
        for (int q = 0; q < world; ++q) sp.p[q] = (const float*)f->post[q].host;
        hipLaunchKernelGGL(fabric_sum_kernel, dim3(1024), dim3(256), 0, stream, n, sp, tmp);
        if (hipStreamSynchronize(stream) != hipSuccess) return fail();
        if (!f->barrier()) return fail();                                 // everybody has read everybody's input
        if (hipMemcpyAsync(dev, tmp, sizeof(float) * n, hipMemcpyDeviceToDevice, stream) != hipSuccess) return fail();
        if (hipStreamSynchronize(stream) != hipSuccess) return fail();
        if (!f->barrier()) return fail();
        return MDX_OK;
    }
    const char* name() const override { return "in-process fabric"; }
    int fail() { f->abort(); FAIL(MDX_EDEVICE, "in-process fabric: a rank failed or aborted"); }
    int exchange(const float4* send, const std::vector<MdxSeg>& ssegs, float4* recv, const std::vector<MdxSeg>& rsegs,
                 hipStream_t stream) override {
        if (dead) FAIL(MDX_EDEVICE, "in-process fabric: an earlier exchange was refused");
        for (const MdxSeg& s : ssegs) if (s.peer < 0 || s.peer >= world) { dead = true; FAIL(MDX_EDEVICE, "in-process fabric: send segment names a peer outside the communicator"); }
        for (const MdxSeg& r : rsegs) if (r.peer < 0 || r.peer >= world) { dead = true; FAIL(MDX_EDEVICE, "in-process fabric: receive segment names a peer outside the communicator"); }
        if (hipStreamSynchronize(stream) != hipSuccess) return fail();   // my rows are complete
        f->post[rank].send = send; f->post[rank].ssegs = &ssegs;
        if (!f->barrier()) return fail();
        for (const MdxSeg& r : rsegs) {
            if (!r.nrows) continue;
            const mdx_fabric::Post& p = f->post[r.peer];
            const MdxSeg* src = nullptr;
            for (const MdxSeg& s : *p.ssegs) if (s.peer == rank) { src = &s; break; }
            if (!src || src->nrows != r.nrows) { f->abort(); FAIL(MDX_EDEVICE, "in-process fabric: send / receive segment mismatch"); }
            if (hipMemcpyAsync(recv + r.row0, p.send + src->row0, sizeof(float4) * r.nrows, hipMemcpyDefault, stream) != hipSuccess) return fail();
        }
        if (hipStreamSynchronize(stream) != hipSuccess) return fail();
        if (!f->barrier()) return fail();                                 // peers may reuse their send buffers
        return MDX_OK;
    }
    int all_reduce(void* dev, size_t n, int kind, hipStream_t stream) override {
        if (n > 4096) FAIL(MDX_EPARAM, "fabric all_reduce is for small arrays");
        const size_t bytes = n * (kind == 0 ? sizeof(double) : sizeof(uint32_t));
        std::vector<unsigned char> mine(bytes), out(bytes);
        if (hipMemcpyAsync(mine.data(), dev, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return fail();
        f->post[rank].host = mine.data();
        if (!f->barrier()) return fail();
        for (size_t i = 0; i < n; ++i) {
            if (kind == 0) { double s = 0.0; for (int q = 0; q < world; ++q) s += ((const double*)f->post[q].host)[i]; ((double*)out.data())[i] = s; }
            else { uint32_t s = 0; for (int q = 0; q < world; ++q) s = std::max(s, ((const uint32_t*)f->post[q].host)[i]); ((uint32_t*)out.data())[i] = s; }
        }
        if (!f->barrier()) return fail();                                 // everybody has read everybody's input
        if (hipMemcpyAsync(dev, out.data(), bytes, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return fail();
        return MDX_OK;
    }
    int all_gather_u32(uint32_t mine, uint32_t* all, hipStream_t stream) override {
        (void)stream;
        f->post[rank].word = mine;
        if (!f->barrier()) return fail();
        for (int q = 0; q < world; ++q) all[q] = f->post[q].word;
        if (!f->barrier()) return fail();
        return MDX_OK;
    }
};

// MDX_NULL_WIRE_US=x: every send/recv group of the null transport occupies its stream for x microseconds (one wave watching the
// 100 MHz wall clock) - the wire time a real exchange would put there, so that what the step hides of it can be measured on one GPU
__global__ void null_wire_kernel(uint32_t us) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(16);
}
struct NullTransport : MdxTransport {
    int wire_us = -1;      // < 0: MDX_NULL_WIRE_US not set (the transport of rounds 3-4: nothing is enqueued, unpack / add kernels are skipped)
    NullTransport() { const char* e = std::getenv("MDX_NULL_WIRE_US"); if (e) wire_us = std::max(0, std::atoi(e)); }
    const char* name() const override { return wire_us >= 0 ? "null (stated wire time, loop-back buffers)" : "null (delivers nothing)"; }
    bool delivers() const override { return false; }
    bool loopback() const override { return wire_us >= 0; }
    bool wire_time_decides() const override { return wire_us >= 0; }
    int exchange(const float4*, const std::vector<MdxSeg>&, float4*, const std::vector<MdxSeg>&, hipStream_t st) override {
        if (wire_us > 0) { hipLaunchKernelGGL(null_wire_kernel, dim3(1), dim3(64), 0, st, (uint32_t)wire_us); HIP_TRY(hipGetLastError()); }
        return MDX_OK;
    }
    int all_reduce(void*, size_t, int, hipStream_t) override { return MDX_OK; }
    int all_reduce_f32(float*, size_t, hipStream_t) override { return MDX_OK; }
    int all_gather_u32(uint32_t mine, uint32_t* all, hipStream_t) override { for (int q = 0; q < world; ++q) all[q] = mine; return MDX_OK; }
};
}  // namespace

// ---- shared-memory transport: ranks are PROCESSES of one host, data is staged through a POSIX shm segment -------------
// Not a performance path (every row crosses the host twice): it lets the complete multi-process flow - one process per
// rank, as the driver launches bench.py - run on a box with ONE GPU, where RCCL refuses two ranks per device, and serves
// hosts without RCCL.  Same interface, same call sequence.
namespace {
struct ShmHeader {
    std::atomic<uint32_t> arrived; std::atomic<uint32_t> generation; std::atomic<uint32_t> aborted; uint32_t world;
    uint64_t slot_bytes;
    std::atomic<uint32_t> magic;     // SHM_READY once rank 0 has initialised everything above (written last, release); SHM_DEAD: a leftover
};
constexpr uint32_t SHM_READY = 0x4D445852u, SHM_DEAD = 0xDEADDEADu;
struct ShmSegEntry { int32_t peer; uint32_t nrows; uint64_t offset; };
struct ShmSlotHead { uint32_t n_entries; uint32_t word; ShmSegEntry e[64]; };

struct ShmTransport : MdxTransport {
    std::string shm_name; int fd = -1; unsigned char* base = nullptr; size_t total = 0; ShmHeader* hd = nullptr; uint64_t slot_bytes = 0;
    bool unlinked = false;
    bool dead = false;
    ~ShmTransport() override {
        if (base) munmap(base, total);
        if (fd >= 0) close(fd);
        if (rank == 0 && !shm_name.empty() && !unlinked) shm_unlink(shm_name.c_str());
    }
    const char* name() const override { return "shared memory (host-staged)"; }
    unsigned char* slot(int q) const { return base + 4096 + (size_t)q * slot_bytes; }
    int fail(const char* why) { if (hd) hd->aborted.store(1); mdx_set_error(std::string("shared-memory transport: ") + why); return MDX_EDEVICE; }
    bool barrier() {
        const uint32_t gen = hd->generation.load();
        if (hd->arrived.fetch_add(1) + 1 == (uint32_t)world) { hd->arrived.store(0); hd->generation.fetch_add(1); return !hd->aborted.load(); }
        const auto t0 = std::chrono::steady_clock::now();
        while (hd->generation.load() == gen) {
            if (hd->aborted.load()) return false;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { hd->aborted.store(1); return false; }
            std::this_thread::yield();
        }
        return !hd->aborted.load();
    }
    int exchange(const float4* send, const std::vector<MdxSeg>& ssegs, float4* recv, const std::vector<MdxSeg>& rsegs, hipStream_t stream) override {
        // a peer outside the communicator is refused before the barrier (nobody would arrive) and before slot() is indexed with it
        if (dead) { mdx_set_error("shared-memory transport: an earlier exchange was refused"); return MDX_EDEVICE; }
        for (const MdxSeg& s : ssegs) if (s.peer < 0 || s.peer >= world) { dead = true; mdx_set_error("shared-memory transport: send segment names a peer outside the communicator"); return MDX_EDEVICE; }
        for (const MdxSeg& r : rsegs) if (r.peer < 0 || r.peer >= world) { dead = true; mdx_set_error("shared-memory transport: receive segment names a peer outside the communicator"); return MDX_EDEVICE; }
        ShmSlotHead* mine = (ShmSlotHead*)slot(rank);
        if (ssegs.size() > 64) return fail("too many segments");
        uint64_t off = sizeof(ShmSlotHead);
        mine->n_entries = (uint32_t)ssegs.size();
        for (size_t k = 0; k < ssegs.size(); ++k) {
            const uint64_t bytes = (uint64_t)ssegs[k].nrows * sizeof(float4);
            if (off + bytes > slot_bytes) return fail("message larger than the shared-memory slot (MDX_SHM_SLOT_MB)");
            mine->e[k] = {ssegs[k].peer, ssegs[k].nrows, off};
            if (bytes && hipMemcpyAsync(slot(rank) + off, send + ssegs[k].row0, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return fail("copy failed");
            off += bytes;
        }
        if (hipStreamSynchronize(stream) != hipSuccess) return fail("sync failed");
        if (!barrier()) return fail("a rank failed or timed out");
        for (const MdxSeg& r : rsegs) {
            if (!r.nrows) continue;
            const ShmSlotHead* ph = (const ShmSlotHead*)slot(r.peer);
            const ShmSegEntry* src = nullptr;
            for (uint32_t k = 0; k < ph->n_entries; ++k) if (ph->e[k].peer == rank) { src = &ph->e[k]; break; }
            if (!src || src->nrows != r.nrows) return fail("send / receive segment mismatch");
            if (hipMemcpyAsync(recv + r.row0, slot(r.peer) + src->offset, (size_t)r.nrows * sizeof(float4), hipMemcpyHostToDevice, stream) != hipSuccess) return fail("copy failed");
        }
        if (hipStreamSynchronize(stream) != hipSuccess) return fail("sync failed");
        if (!barrier()) return fail("a rank failed or timed out");
        return MDX_OK;
    }
    template <typename T, typename F>
    int reduce_impl(T* dev, size_t n, hipStream_t stream, F op) {
        const size_t bytes = n * sizeof(T);
        if (sizeof(ShmSlotHead) + bytes > slot_bytes) return fail("array larger than the shared-memory slot (MDX_SHM_SLOT_MB)");
        T* mine = (T*)(slot(rank) + sizeof(ShmSlotHead));
        if (hipMemcpyAsync(mine, dev, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return fail("copy failed");
        if (!barrier()) return fail("a rank failed or timed out");
        std::vector<T> out(n);
        for (size_t i = 0; i < n; ++i) {
            T a = ((const T*)(slot(0) + sizeof(ShmSlotHead)))[i];
            for (int q = 1; q < world; ++q) a = op(a, ((const T*)(slot(q) + sizeof(ShmSlotHead)))[i]);     // rank order: same bits everywhere
            out[i] = a;
        }
        if (!barrier()) return fail("a rank failed or timed out");
        if (hipMemcpyAsync(dev, out.data(), bytes, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return fail("copy failed");
        return MDX_OK;
    }
    int all_reduce(void* dev, size_t n, int kind, hipStream_t stream) override {
        if (kind == 0) return reduce_impl((double*)dev, n, stream, [](double a, double b) { return a + b; });
        return reduce_impl((uint32_t*)dev, n, stream, [](uint32_t a, uint32_t b) { return std::max(a, b); });
    }
    int all_reduce_f32(float* dev, size_t n, hipStream_t stream) override { return reduce_impl(dev, n, stream, [](float a, float b) { return a + b; }); }
    int all_gather_u32(uint32_t mine, uint32_t* all, hipStream_t) override {
        ((ShmSlotHead*)slot(rank))->word = mine;
        if (!barrier()) return fail("a rank failed or timed out");
        for (int q = 0; q < world; ++q) all[q] = ((const ShmSlotHead*)slot(q))->word;
        if (!barrier()) return fail("a rank failed or timed out");
        return MDX_OK;
    }
};
}  // namespace

MdxTransport* mdx_make_shm_transport(const char* name, int rank, int world) {
    ShmTransport* t = new ShmTransport();
    t->rank = rank; t->world = world; t->shm_name = std::string("/mdx_") + name;
    size_t slot_mb = 64;
    if (const char* e = std::getenv("MDX_SHM_SLOT_MB")) slot_mb = (size_t)std::max(1, std::atoi(e));
    t->slot_bytes = slot_mb << 20;
    t->total = 4096 + (size_t)world * t->slot_bytes;
    // Rank 0 creates the segment afresh: a leftover of the same name (a crashed run, a reused pid in a test's name) is first
    // poisoned - a rank that attached to it fails its rendezvous instead of waiting on stale counters - then unlinked, and
    // the new one is opened O_EXCL.  Readiness is ONE word written last (release); the others take nothing from the header
    // before they have seen it (acquire).  Once every rank has mapped it the name is unlinked: the mappings stay valid and
    // nothing is left in /dev/shm if the run dies later (round-2 advisor finding).
    const auto t0 = std::chrono::steady_clock::now();
    auto expired = [&](int secs) { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(secs); };
    auto unmap = [&]() { if (t->base) munmap(t->base, t->total); t->base = nullptr; t->hd = nullptr; if (t->fd >= 0) close(t->fd); t->fd = -1; };
    if (rank == 0) {
        int old = shm_open(t->shm_name.c_str(), O_RDWR, 0600);
        if (old >= 0) {
            struct stat st;
            if (fstat(old, &st) == 0 && (size_t)st.st_size >= sizeof(ShmHeader)) {
                void* om = mmap(nullptr, sizeof(ShmHeader), PROT_READ | PROT_WRITE, MAP_SHARED, old, 0);
                if (om != MAP_FAILED) { ShmHeader* oh = (ShmHeader*)om; oh->magic.store(SHM_DEAD); oh->aborted.store(1); munmap(om, sizeof(ShmHeader)); }
            }
            close(old);
            shm_unlink(t->shm_name.c_str());
        }
        t->fd = shm_open(t->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (t->fd < 0) { mdx_set_error("shared-memory transport: cannot create the segment (another run with the same name?)"); delete t; return nullptr; }
        if (ftruncate(t->fd, (off_t)t->total) != 0) { mdx_set_error("shared-memory transport: ftruncate failed"); shm_unlink(t->shm_name.c_str()); delete t; return nullptr; }
        void* m = mmap(nullptr, t->total, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
        if (m == MAP_FAILED) { mdx_set_error("shared-memory transport: mmap failed"); shm_unlink(t->shm_name.c_str()); delete t; return nullptr; }
        t->base = (unsigned char*)m; t->hd = (ShmHeader*)m;
        t->hd->arrived.store(0); t->hd->generation.store(0); t->hd->aborted.store(0); t->hd->slot_bytes = t->slot_bytes; t->hd->world = (uint32_t)world;
        t->hd->magic.store(SHM_READY, std::memory_order_release);
        if (!t->barrier()) { mdx_set_error("shared-memory transport: rendezvous failed"); shm_unlink(t->shm_name.c_str()); t->unlinked = true; delete t; return nullptr; }
        shm_unlink(t->shm_name.c_str()); t->unlinked = true;       // every rank has it mapped
        return t;
    }
    for (;;) {      // ranks > 0: attach to a READY segment of the right shape; a poisoned or half-built one is tried again
        if (expired(60)) { mdx_set_error("shared-memory transport: the segment never appeared"); delete t; return nullptr; }
        t->fd = shm_open(t->shm_name.c_str(), O_RDWR, 0600);
        struct stat st;
        if (t->fd < 0 || fstat(t->fd, &st) != 0 || (size_t)st.st_size < t->total) { unmap(); std::this_thread::sleep_for(std::chrono::milliseconds(5)); continue; }
        void* m = mmap(nullptr, t->total, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
        if (m == MAP_FAILED) { mdx_set_error("shared-memory transport: mmap failed"); delete t; return nullptr; }
        t->base = (unsigned char*)m; t->hd = (ShmHeader*)m;
        bool ready = false;
        for (int spin = 0; spin < 400 && !ready; ++spin) {          // up to ~2 s for rank 0 to finish this segment's header
            const uint32_t mg = t->hd->magic.load(std::memory_order_acquire);
            if (mg == SHM_READY) ready = true;
            else if (mg == SHM_DEAD) break;
            else std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
        if (ready && t->hd->world == (uint32_t)world && t->hd->slot_bytes == t->slot_bytes && t->barrier()) return t;
        unmap();                                                     // a leftover that rank 0 has poisoned (or is about to): look again
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

MdxTransport* mdx_make_fabric_transport(mdx_fabric* f, int rank) {
    if (!f || rank < 0 || rank >= f->world) { mdx_set_error("bad fabric / rank"); return nullptr; }
    FabricTransport* t = new FabricTransport();
    t->f = f; t->rank = rank; t->world = f->world;
    return t;
}
MdxTransport* mdx_make_null_transport(int rank, int world) {
    NullTransport* t = new NullTransport();
    t->rank = rank; t->world = world;
    return t;
}

// ---- C ABI: attach a transport to a handle -----------------------------------------------------------------------
extern "C" int mdx_comm_init(mdx_handle* h, const uint8_t id[128], int rank, int world) {
    if (!h || !id) FAIL(MDX_EPARAM, "null argument");
    if (world < 1 || world > 32 || rank < 0 || rank >= world) FAIL(MDX_EPARAM, "rank / world out of range (world <= 32)");
    if (h->dd) FAIL(MDX_EPARAM, "the handle is already decomposed");
    MdxTransport* t = mdx_make_rccl_transport(id, rank, world, h->device);
    if (!t) return MDX_EDEVICE;
    return mdx_dd_attach(h, t);
}
extern "C" int mdx_comm_init_fabric(mdx_handle* h, mdx_fabric* f, int rank) {
    if (!h || !f) FAIL(MDX_EPARAM, "null argument");
    if (h->dd) FAIL(MDX_EPARAM, "the handle is already decomposed");
    if (f->world > 32) FAIL(MDX_EPARAM, "world <= 32");
    MdxTransport* t = mdx_make_fabric_transport(f, rank);
    if (!t) return MDX_EPARAM;
    const int rc = mdx_dd_attach(h, t);
    if (rc != MDX_OK) f->abort();
    return rc;
}
extern "C" int mdx_comm_init_shm(mdx_handle* h, const char* name, int rank, int world) {
    if (!h || !name || !name[0]) FAIL(MDX_EPARAM, "null argument");
    if (world < 1 || world > 32 || rank < 0 || rank >= world) FAIL(MDX_EPARAM, "rank / world out of range (world <= 32)");
    if (h->dd) FAIL(MDX_EPARAM, "the handle is already decomposed");
    MdxTransport* t = mdx_make_shm_transport(name, rank, world);
    if (!t) return MDX_EDEVICE;
    return mdx_dd_attach(h, t);
}
extern "C" int mdx_comm_init_null(mdx_handle* h, int rank, int world) {
    if (!h) FAIL(MDX_EPARAM, "null argument");
    if (world < 1 || world > 32 || rank < 0 || rank >= world) FAIL(MDX_EPARAM, "rank / world out of range");
    if (h->dd) FAIL(MDX_EPARAM, "the handle is already decomposed");
    return mdx_dd_attach(h, mdx_make_null_transport(rank, world));
}
