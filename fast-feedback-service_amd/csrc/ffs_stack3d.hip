// ffs_stack3d.hip -- rotation sweeps: the per-frame strong-pixel lists never leave the GPU; they are appended to the
// stack's device buffers and ffs_stack3d_finish labels the whole stack in 3D (kernels_stack3d.hpp).  Also the one
// exchange step of a multi-GPU sweep (lists -> the GPU that owns the stack) and ffs_multi_init.
// Reference: ConnectedComponents::find_3d_components, spotfinder/connected_components/connected_components.cc:270-470;
// the driver's rotation_slices map, spotfinder/spotfinder.cc:913-918,1099-1148.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is loaded with dlopen when several devices are in use

#include "ffs_internal.hpp"
#include "kernels_stack3d.hpp"

// Growable device (or pinned host) buffer of the 3D stack: reallocated with slack when too small
template <typename T>
struct PoolBuf {
    T* p = nullptr;
    size_t cap = 0;
    bool pinned_host = false;
    hipError_t ensure(size_t n, bool keep = false, hipStream_t st = nullptr) {
        if (n <= cap) return hipSuccess;
        const size_t want = std::max<size_t>(n + n / 2, 1024);
        T* q = nullptr;
        hipError_t e = pinned_host ? hipHostMalloc(reinterpret_cast<void**>(&q), want * sizeof(T), hipHostMallocDefault)
                                   : hipMalloc(reinterpret_cast<void**>(&q), want * sizeof(T) + 256);
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(q, p, cap * sizeof(T), pinned_host ? hipMemcpyHostToHost : hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { pinned_host ? (void)hipHostFree(q) : (void)hipFree(q); return e; }
        }
        release();
        p = q;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) pinned_host ? (void)hipHostFree(p) : (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ffs_stack3d {
    ffs_ctx* ctx = nullptr;
    uint64_t max_total = 0;
    hipStream_t st = nullptr;  // the stack's own stream
    std::mutex mu;             // worker threads add their batches concurrently (the reference's rotation_slices_mutex)
    // slices as they arrive: lists appended to the arrival buffers on the device
    struct Slice {
        uint32_t off = 0, n = 0;
    };
    std::map<int64_t, Slice> slices;  // frame id -> place in the arrival buffers (std::map order = z order, spotfinder.cc:1105-1108)
    uint64_t arrived = 0;
    PoolBuf<uint32_t> a_k, a_i;
    PoolBuf<StackSlice> d_table, h_table;
    // ffs_stack3d_add_batch does not wait for its append: each call takes the next of kSlots slice tables and leaves an
    // event behind; a slot is reused only after its event, and finish / a growing buffer / destroy wait for all of them
    static constexpr int kSlots = 8;
    PoolBuf<StackSlice> d_ring, h_ring;   // kSlots tables of ring_stride entries
    size_t ring_stride = 0;
    hipEvent_t slot_ev[kSlots] = {};
    bool slot_used[kSlots] = {};
    uint64_t n_adds = 0;
    // the stack in (z, k) order and the scratch of its labelling (sized in finish, kept for the next one)
    PoolBuf<uint32_t> d_k, d_i, d_z, d_parent, d_comp, d_begin, d_chunk_roots, d_small, d_sx, d_sy, d_sc;
    PoolBuf<CompAcc> d_acc;
    PoolBuf<ReflOut> d_recs;
    std::vector<ffs_reflection> out;
    // per-signal view of the last finish (vertex order)
    std::vector<uint32_t> sig_x, sig_y, sig_i;
    std::vector<int32_t> sig_z, sig_refl;
    float last_finish_ms = 0;
};

// ---- 3D stack ------------------------------------------------------------------------------------------

static int stack3d_create_impl(ffs_ctx* c, uint64_t max_total, ffs_stack3d** out) {
    {   // a stack of this context that was destroyed: its stream and its (grown) buffers are ready -- a sweep's worth of
        // hipMalloc / hipFree is 1.5 ms, more than its 100 frames take on the GPU
        std::lock_guard<std::mutex> lock(c->stream_mu);
        if (!c->stack_pool.empty()) {
            ffs_stack3d* st = c->stack_pool.back();
            c->stack_pool.pop_back();
            st->max_total = max_total ? max_total : (1ull << 30);
            *out = st;
            return FFS_OK;
        }
    }
    ffs_stack3d* st = new (std::nothrow) ffs_stack3d();
    if (!st) return FFS_ERR_NOMEM;
    st->ctx = c;
    st->max_total = max_total ? max_total : (1ull << 30);
    st->h_table.pinned_host = true;
    HIP_TRY(c, hipSetDevice(c->device));
    hipError_t e = hipStreamCreateWithFlags(&st->st, hipStreamNonBlocking);
    if (e != hipSuccess) {
        c->err = std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e);
        delete st;
        return FFS_ERR_DEVICE;
    }
    *out = st;
    return FFS_OK;
}

std::atomic<int> g_live_stacks{0};   // while one is alive, batches leave their strong-pixel lists on the device (tuning "device_lists" = 2)

extern "C" int ffs_stack3d_create(ffs_ctx* c, uint64_t max_total, ffs_stack3d** out) {
    if (!c || !out) return FFS_ERR_INVALID;
    *out = nullptr;
    const int rc = stack3d_create_impl(c, max_total, out);
    if (rc != FFS_OK) return rc;
    g_live_stacks.fetch_add(1);
    {
        std::lock_guard<std::mutex> lock(c->stream_mu);
        c->live_stacks.push_back(*out);
    }
    handle_add(kHandleStack, *out);
    return FFS_OK;
}

extern "C" void ffs_stack3d_destroy(ffs_stack3d* st) {
    if (!st || !handle_take(kHandleStack, st)) return;   // (destroyed already -- by its context's ffs_ctx_destroy, or twice)
    g_live_stacks.fetch_sub(1);
    ffs_ctx* c = st->ctx;   // alive: a context takes its stacks with it, and this one was still in the registry
    (void)hipSetDevice(c->device);
    {
        std::lock_guard<std::mutex> lock(c->stream_mu);
        auto& v = c->live_stacks;
        v.erase(std::remove(v.begin(), v.end(), st), v.end());
        if (c->stack_pool.size() < 2) {   // keep it, emptied, for the next sweep
            for (int k = 0; k < ffs_stack3d::kSlots; ++k)
                if (st->slot_used[k]) {
                    (void)hipEventSynchronize(st->slot_ev[k]);
                    st->slot_used[k] = false;
                }
            if (st->st) (void)hipStreamSynchronize(st->st);
            st->slices.clear();
            st->arrived = 0;
            st->n_adds = 0;
            st->out.clear();
            st->sig_x.clear(); st->sig_y.clear(); st->sig_z.clear(); st->sig_i.clear(); st->sig_refl.clear();
            st->last_finish_ms = 0;
            c->stack_pool.push_back(st);
            return;
        }
    }
    stack3d_free(st);
}

void stack3d_free(ffs_stack3d* st) {
    (void)hipSetDevice(st->ctx->device);
    for (int k = 0; k < ffs_stack3d::kSlots; ++k)
        if (st->slot_ev[k]) {
            if (st->slot_used[k]) (void)hipEventSynchronize(st->slot_ev[k]);
            (void)hipEventDestroy(st->slot_ev[k]);
        }
    if (st->st) (void)hipStreamSynchronize(st->st);
    st->d_ring.release(); st->h_ring.release();
    st->a_k.release(); st->a_i.release(); st->d_table.release(); st->h_table.release();
    st->d_k.release(); st->d_i.release(); st->d_z.release(); st->d_parent.release(); st->d_comp.release();
    st->d_begin.release(); st->d_chunk_roots.release(); st->d_small.release();
    st->d_sx.release(); st->d_sy.release(); st->d_sc.release(); st->d_acc.release(); st->d_recs.release();
    if (st->st) (void)hipStreamDestroy(st->st);
    delete st;
}

#define STK_TRY(c, expr)                                                        \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);       \
            return e_ == hipErrorOutOfMemory ? FFS_ERR_NOMEM : FFS_ERR_DEVICE;  \
        }                                                                       \
    } while (0)

// the appends still in flight (ffs_stack3d_add_batch leaves them running) are done when this returns
static void stack3d_join_appends(ffs_stack3d* st) {
    for (int k = 0; k < ffs_stack3d::kSlots; ++k)
        if (st->slot_used[k]) {
            (void)hipEventSynchronize(st->slot_ev[k]);
            st->slot_used[k] = false;
        }
}

// room for `more` entries behind the ones that have arrived (the lists already there are kept)
static int stack3d_reserve(ffs_stack3d* st, uint64_t more) {
    ffs_ctx* c = st->ctx;
    if (st->arrived + more > st->max_total || st->arrived + more >= (1ull << 32) - 1) {
        c->err = "ffs_stack3d: too many strong pixels in the stack";
        return FFS_ERR_OVERFLOW;
    }
    if (st->arrived + more > st->a_k.cap || st->arrived + more > st->a_i.cap) stack3d_join_appends(st);  // (the buffers move)
    STK_TRY(c, st->a_k.ensure(st->arrived + more, true, st->st));
    STK_TRY(c, st->a_i.ensure(st->arrived + more, true, st->st));
    return FFS_OK;
}

static int stack3d_add_slice_impl(ffs_stack3d* st, int64_t frame_id, const uint32_t* k, const uint32_t* inten, uint32_t n) {
    ffs_ctx* c = st->ctx;
    std::lock_guard<std::mutex> lock(st->mu);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = stack3d_reserve(st, n);
    if (rc != FFS_OK) return rc;
    if (n) {
        STK_TRY(c, hipMemcpyAsync(st->a_k.p + st->arrived, k, (size_t)n * 4, hipMemcpyHostToDevice, st->st));
        STK_TRY(c, hipMemcpyAsync(st->a_i.p + st->arrived, inten, (size_t)n * 4, hipMemcpyHostToDevice, st->st));
        STK_TRY(c, hipStreamSynchronize(st->st));  // the caller's arrays may go away
    }
    st->slices[frame_id] = ffs_stack3d::Slice{(uint32_t)st->arrived, n};  // (a frame added twice: the later list counts)
    st->arrived += n;
    return FFS_OK;
}

extern "C" int ffs_stack3d_add_slice(ffs_stack3d* st, int64_t frame_id, const uint32_t* k,
                                     const uint32_t* inten, uint32_t n) {
    if (!st || (n && (!k || !inten))) return FFS_ERR_INVALID;
    return guarded(st->ctx, [&] { return stack3d_add_slice_impl(st, frame_id, k, inten, n); });
}

// ---- several GPUs in one process: the exchange step of rotation sweeps ---------------------------------------
// Frames are independent, so a driver with one context per GPU needs no collective for stills.  A rotation
// sweep does have one exchange: every frame's strong-pixel list has to reach the GPU that owns the 3D stack.
// Transport between two different devices: RCCL point-to-point (ncclSend / ncclRecv inside one group, over
// xGMI) when librccl can be loaded and ffs_multi_init() built the communicators, else hipMemcpyPeerAsync.
// The reference has nothing to compare with: one process, one device (src/ffs/cuda_arg_parser.cc:56-61).
//
// Locking: ffs_multi_init() writes this state under g_multi_mu and must not run while batches are being added;
// the transfers only read it.  What serialises the use of the home rank's communicator is the stack's own mutex
// (every transfer into a stack holds it), and nothing in a transfer waits on the host: the receive (or peer copy)
// is ordered on the stack's stream, which ffs_stack3d_finish uses too.
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static std::mutex g_multi_mu;
static RcclApi g_rccl;
static std::vector<int> g_comm_devices;     // distinct devices, rank = position
static std::vector<ncclComm_t> g_comms;     // one communicator per rank (ncclCommInitAll)
enum { kTransportNone = 0, kTransportPeer = 1, kTransportRccl = 2 };
static std::atomic<int> g_multi_transport{kTransportNone};
static bool g_want_rccl = true;             // FFS_GATHER / the transport argument, as given to ffs_multi_init
static bool g_gather_forced = false;        // ... explicitly (then RCCL is used even between contexts on one GPU)

static int comm_rank_of(int device) {
    for (size_t r = 0; r < g_comm_devices.size(); ++r)
        if (g_comm_devices[r] == device) return (int)r;
    return -1;
}

extern "C" int ffs_multi_init(const int* devices, int n_devices, const char* transport) {
    if (!devices || n_devices <= 0) return FFS_ERR_INVALID;
    std::lock_guard<std::mutex> lock(g_multi_mu);
    std::vector<int> distinct;
    for (int i = 0; i < n_devices; ++i)
        if (std::find(distinct.begin(), distinct.end(), devices[i]) == distinct.end()) distinct.push_back(devices[i]);
    const char* env = std::getenv("FFS_GATHER");   // "rccl" | "peer": how rotation lists travel (results are the same)
    const std::string want = transport ? transport : (env ? env : "rccl");
    g_gather_forced = transport != nullptr || env != nullptr;
    g_want_rccl = want == "rccl";
    if (!g_comms.empty() && distinct == g_comm_devices) {   // the communicators are there already: only the choice of transport may change
        g_multi_transport = g_want_rccl ? kTransportRccl : (distinct.size() > 1 ? kTransportPeer : kTransportNone);
        return FFS_OK;
    }
    if (!g_comms.empty() && g_rccl.CommDestroy) {
        for (ncclComm_t cm : g_comms) (void)g_rccl.CommDestroy(cm);
        g_comms.clear();
    }
    g_comm_devices = distinct;
    g_multi_transport = distinct.size() > 1 ? kTransportPeer : kTransportNone;
    if (!g_want_rccl) return FFS_OK;
    if (!g_rccl.lib) {
        g_rccl.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl.lib) g_rccl.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) {
            g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(dlsym(g_rccl.lib, "ncclCommInitAll"));
            g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.lib, "ncclCommDestroy"));
            g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(dlsym(g_rccl.lib, "ncclGroupStart"));
            g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(dlsym(g_rccl.lib, "ncclGroupEnd"));
            g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(dlsym(g_rccl.lib, "ncclSend"));
            g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(dlsym(g_rccl.lib, "ncclRecv"));
            g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.lib, "ncclAllGather"));
            g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.lib, "ncclGetErrorString"));
        }
    }
    if (!g_rccl.lib || !g_rccl.CommInitAll || !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return FFS_OK;  // no RCCL here: peer copies
    g_comms.assign(distinct.size(), nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(g_comms.data(), (int)distinct.size(), distinct.data());
    if (r != ncclSuccess) {
        g_comms.clear();
        g_create_error = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
        return FFS_OK;  // still usable through peer copies; ffs_multi_transport() tells which
    }
    g_multi_transport = kTransportRccl;
    return FFS_OK;
}

extern "C" const char* ffs_multi_transport(void) {   // (string literals: nothing a later ffs_multi_init could invalidate)
    switch (g_multi_transport.load()) {
    case kTransportRccl: return "rccl";
    case kTransportPeer: return "peer";
    default: return "none";
    }
}

// NUMA node of a GPU (sysfs: /sys/bus/pci/devices/<bus id>/numa_node), -1 if unknown: where the driver should keep the
// worker threads that feed it, and their pinned buffers
extern "C" int ffs_device_numa_node(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* p = bus; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

// ---- the gather of the per-frame spot lists over RCCL (BASELINE.json north_star) -------------------------
// One process, one context per GPU: after a batch each context holds its reflections' centre rows (frame id, x, y, z) on the host
// (ffs_stream_spot_centres).  The driver can simply read them there (`spotfinder --gather host`); this is the collective the
// north star names, on the communicators ffs_multi_init built: every rank's row count by ncclAllGather, then exactly the written
// rows by ncclSend / ncclRecv inside one group to the root's device, and one copy from there to the host.  Contexts that share a
// GPU share its rank: their rows go out as ONE message (one send per pair of ranks and group: two sends between the same pair were
// matched out of order with their receives in the one-GPU rehearsal).  Scratch per rank's first stream, allocated on first use, freed with it.
struct GatherScratch {
    float* d_rows = nullptr;            // this stream's rows on its device
    size_t rows_cap = 0;
    unsigned long long* d_cnt = nullptr;   // [1] this rank's count | [ranks] everybody's (behind it)
    float* d_all = nullptr;             // root only: every rank's rows, in stream order
    size_t all_cap = 0;
    hipStream_t st = nullptr;
};
static std::mutex g_gather_mu;
static std::map<ffs_stream*, GatherScratch> g_gather;   // (a handful of entries: one per stream that ever took part)

void gather_scratch_free(ffs_stream* s) {
    std::lock_guard<std::mutex> lock(g_gather_mu);
    auto it = g_gather.find(s);
    if (it == g_gather.end()) return;
    GatherScratch& g = it->second;
    if (g.st) (void)hipStreamSynchronize(g.st);
    if (g.d_rows) (void)hipFree(g.d_rows);
    if (g.d_cnt) (void)hipFree(g.d_cnt);
    if (g.d_all) (void)hipFree(g.d_all);
    if (g.st) (void)hipStreamDestroy(g.st);
    g_gather.erase(it);
}

static int multi_gather_rows_impl(ffs_stream* const* streams, uint32_t n, uint32_t root, float* rows4_out, uint32_t cap, uint32_t* n_rows) {
    ffs_ctx* rc = streams[root]->ctx;
    std::lock_guard<std::mutex> lock_multi(g_multi_mu);   // (the communicators are not ours alone: a rotation exchange may want them too)
    if (g_comms.empty() || !g_rccl.AllGather || !g_rccl.Send || !g_rccl.Recv) {
        rc->err = "ffs_multi_gather_rows: no RCCL communicators (ffs_multi_init with transport \"rccl\" first; ffs_multi_transport() tells)";
        return FFS_ERR_INVALID;
    }
    const int n_ranks = (int)g_comms.size();
    std::vector<int> rank(n), leader((size_t)n_ranks, -1);
    std::vector<uint64_t> rows(n), per_rank((size_t)n_ranks, 0), in_rank(n, 0), rank_at((size_t)n_ranks + 1, 0);
    std::lock_guard<std::mutex> lock(g_gather_mu);
    for (uint32_t i = 0; i < n; ++i) {
        ffs_stream* s = streams[i];
        if (s->busy) { rc->err = "ffs_multi_gather_rows: a stream has a batch in flight (call after ffs_wait)"; return FFS_ERR_INVALID; }
        rank[i] = comm_rank_of(s->ctx->device);
        if (rank[i] < 0) { rc->err = "ffs_multi_gather_rows: a stream's device is not in the communicator"; return FFS_ERR_INVALID; }
        if (leader[(size_t)rank[i]] < 0) leader[(size_t)rank[i]] = (int)i;   // a rank's first stream holds its send buffer and its count
        rows[i] = s->centres.size() / 4;
        in_rank[i] = per_rank[(size_t)rank[i]];                             // where this stream's rows start inside its rank's message
        per_rank[(size_t)rank[i]] += rows[i];
    }
    for (int r = 0; r < n_ranks; ++r) {
        if (leader[(size_t)r] < 0) { rc->err = "ffs_multi_gather_rows: every rank of the communicator needs a stream (a collective)"; return FFS_ERR_INVALID; }
        rank_at[(size_t)r + 1] = rank_at[(size_t)r] + per_rank[(size_t)r];
    }
    const uint64_t total = rank_at[(size_t)n_ranks];
    if (n_rows) *n_rows = (uint32_t)std::min<uint64_t>(total, 0xFFFFFFFFull);
    if (total > cap) { rc->err = "ffs_multi_gather_rows: the rows do not fit `cap`"; return FFS_ERR_OVERFLOW; }
    const int root_rank = rank[root];
    // scratch of every rank's first stream: one message per rank -- the rows of all its contexts, one behind the other -- and the count
    for (int r = 0; r < n_ranks; ++r) {
        ffs_stream* s = streams[(size_t)leader[(size_t)r]];
        GatherScratch& g = g_gather[s];
        STK_TRY(rc, hipSetDevice(s->ctx->device));
        if (!g.st) STK_TRY(rc, hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking));
        if (!g.d_cnt) STK_TRY(rc, hipMalloc(reinterpret_cast<void**>(&g.d_cnt), (size_t)(1 + n_ranks) * 8 + 256));
        if (per_rank[(size_t)r] > g.rows_cap) {
            STK_TRY(rc, hipStreamSynchronize(g.st));
            if (g.d_rows) (void)hipFree(g.d_rows);
            g.d_rows = nullptr;
            g.rows_cap = per_rank[(size_t)r] + per_rank[(size_t)r] / 2 + 1024;
            STK_TRY(rc, hipMalloc(reinterpret_cast<void**>(&g.d_rows), g.rows_cap * 16));
        }
        if (r == root_rank && total > g.all_cap) {
            STK_TRY(rc, hipStreamSynchronize(g.st));
            if (g.d_all) (void)hipFree(g.d_all);
            g.d_all = nullptr;
            g.all_cap = total + total / 2 + 1024;
            STK_TRY(rc, hipMalloc(reinterpret_cast<void**>(&g.d_all), g.all_cap * 16));
        }
        STK_TRY(rc, hipMemcpyAsync(g.d_cnt, &per_rank[(size_t)r], 8, hipMemcpyHostToDevice, g.st));
    }
    for (uint32_t i = 0; i < n; ++i) {   // every stream's rows onto its rank's device, into the rank's message
        if (!rows[i]) continue;
        GatherScratch& g = g_gather[streams[(size_t)leader[(size_t)rank[i]]]];
        STK_TRY(rc, hipSetDevice(streams[i]->ctx->device));
        STK_TRY(rc, hipMemcpyAsync(g.d_rows + in_rank[i] * 4, streams[i]->centres.data(), rows[i] * 16, hipMemcpyHostToDevice, g.st));
    }
    auto nccl_fail = [&](const char* what, ncclResult_t r) {
        rc->err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
        return FFS_ERR_DEVICE;
    };
    {   // counts
        ncclResult_t r = g_rccl.GroupStart();
        for (int k = 0; k < n_ranks && r == ncclSuccess; ++k) {
            ffs_stream* s = streams[(size_t)leader[(size_t)k]];
            GatherScratch& g = g_gather[s];
            (void)hipSetDevice(s->ctx->device);
            r = g_rccl.AllGather(g.d_cnt, g.d_cnt + 1, 1, ncclUint64, g_comms[(size_t)k], g.st);
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r != ncclSuccess || re != ncclSuccess) return nccl_fail("ncclAllGather of the row counts", r != ncclSuccess ? r : re);
    }
    // rows: exactly what was written, one message per rank, to the root's device.  The root's own message is a device copy; with the
    // transport forced to RCCL (ffs_multi_init(.., "rccl") / FFS_GATHER) it is sent to its own rank, which a one-GPU box can rehearse.
    GatherScratch& gr = g_gather[streams[(size_t)leader[(size_t)root_rank]]];
    {
        ncclResult_t r = g_rccl.GroupStart();
        for (int k = 0; k < n_ranks && r == ncclSuccess; ++k) {
            if (!per_rank[(size_t)k] || (k == root_rank && !g_gather_forced)) continue;
            ffs_stream* s = streams[(size_t)leader[(size_t)k]];
            GatherScratch& g = g_gather[s];
            (void)hipSetDevice(s->ctx->device);
            r = g_rccl.Send(g.d_rows, per_rank[(size_t)k] * 4, ncclFloat, root_rank, g_comms[(size_t)k], g.st);
            (void)hipSetDevice(rc->device);
            if (r == ncclSuccess) r = g_rccl.Recv(gr.d_all + rank_at[(size_t)k] * 4, per_rank[(size_t)k] * 4, ncclFloat, k, g_comms[(size_t)root_rank], gr.st);
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r != ncclSuccess || re != ncclSuccess) return nccl_fail("ncclSend / ncclRecv of the spot rows", r != ncclSuccess ? r : re);
    }
    STK_TRY(rc, hipSetDevice(rc->device));
    if (per_rank[(size_t)root_rank] && !g_gather_forced)
        STK_TRY(rc, hipMemcpyAsync(gr.d_all + rank_at[(size_t)root_rank] * 4, gr.d_rows, per_rank[(size_t)root_rank] * 16, hipMemcpyDeviceToDevice, gr.st));
    std::vector<unsigned long long> counts((size_t)n_ranks, 0);
    STK_TRY(rc, hipMemcpyAsync(counts.data(), gr.d_cnt + 1, (size_t)n_ranks * 8, hipMemcpyDeviceToHost, gr.st));
    if (total) STK_TRY(rc, hipMemcpyAsync(rows4_out, gr.d_all, total * 16, hipMemcpyDeviceToHost, gr.st));
    for (int r = 0; r < n_ranks; ++r) {   // (the senders' streams too: their scratch is reused by the next gather)
        ffs_stream* s = streams[(size_t)leader[(size_t)r]];
        STK_TRY(rc, hipSetDevice(s->ctx->device));
        STK_TRY(rc, hipStreamSynchronize(g_gather[s].st));
    }
    for (int r = 0; r < n_ranks; ++r)
        if (counts[(size_t)r] != per_rank[(size_t)r]) {
            rc->err = "ffs_multi_gather_rows: the gathered counts differ from what the ranks sent";
            return FFS_ERR_DEVICE;
        }
    return FFS_OK;
}

extern "C" int ffs_multi_gather_rows(ffs_stream* const* streams, uint32_t n_streams, uint32_t root, float* rows4_out, uint32_t cap, uint32_t* n_rows) {
    if (!streams || n_streams == 0 || root >= n_streams || !rows4_out) return FFS_ERR_INVALID;
    for (uint32_t i = 0; i < n_streams; ++i)
        if (!streams[i] || !handle_live(kHandleStream, streams[i])) return FFS_ERR_INVALID;
    return guarded(streams[root]->ctx, [&] { return multi_gather_rows_impl(streams, n_streams, root, rows4_out, cap, n_rows); });
}

static const OverflowFrame* overflow_of(const ffs_stream* s, uint32_t f) {
    for (const OverflowFrame& q : s->ovf)
        if (q.frame == f) return &q;
    return nullptr;
}

// Lists of a batch processed on ANOTHER context (same detector geometry, usually another GPU) into this stack: packed end
// to end on the source device, then one transfer per array.  Frames that overflowed the stream's lists were re-run on the
// stream's one-frame stream and have their lists on the host (as for a local batch): those go up from there.
// `tab`: per frame (offset from 0 in the pack, entries packed: 0 for an overflow frame).  Caller holds st->mu.
static int stack3d_add_batch_remote(ffs_stack3d* st, ffs_stream* s, uint64_t packed, uint32_t biggest) {
    ffs_ctx* c = st->ctx;       // home
    ffs_ctx* sc = s->ctx;       // source
    const uint32_t nf = s->n_frames;
    HIP_TRY(c, hipSetDevice(sc->device));
    if (!s->d_pack_k) {
        const size_t cap_all = (size_t)s->max_batch * s->cap;
        if (dmalloc(&s->d_pack_k, cap_all * 4) != hipSuccess || dmalloc(&s->d_pack_i, cap_all * 4) != hipSuccess
            || dmalloc(&s->d_pack_tab, (size_t)s->max_batch * sizeof(StackSlice)) != hipSuccess
            || hipHostMalloc(reinterpret_cast<void**>(&s->h_pack_tab), (size_t)s->max_batch * sizeof(StackSlice), hipHostMallocDefault) != hipSuccess
            || hipEventCreateWithFlags(&s->ev_pack, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            c->err = "allocation of the list pack buffers failed";
            return FFS_ERR_NOMEM;
        }
    }
    if (packed == 0) return FFS_OK;
    // the pack buffers are free again once the previous batch's transfer out of them has completed (peer copies run in the
    // home stream; RCCL sends in this very stream)
    if (s->sent_pending) {
        STK_TRY(c, hipStreamWaitEvent(s->st2, s->ev_sent, 0));
        s->sent_pending = false;
    }
    uint32_t at = 0;
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t n = overflow_of(s, f) ? 0u : s->results[f].num_strong_pixels;
        s->h_pack_tab[f] = StackSlice{0u, at, n, 0u};
        at += n;
    }
    STK_TRY(c, hipMemcpyAsync(s->d_pack_tab, s->h_pack_tab, (size_t)nf * sizeof(StackSlice), hipMemcpyHostToDevice, s->st2));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_stack_append, dim3(std::min<uint32_t>(64, (biggest + 255) / 256), nf), dim3(256), 0, s->st2,
                       s->d_list_k, s->d_list_i, (uint64_t)s->cap, s->d_pack_tab, s->d_pack_k, s->d_pack_i);
    STK_TRY(c, hipGetLastError());
    uint32_t* dst_k = st->a_k.p + st->arrived;
    uint32_t* dst_i = st->a_i.p + st->arrived;
    const int r_src = comm_rank_of(sc->device), r_home = comm_rank_of(c->device);
    const bool use_rccl = !g_comms.empty() && r_src >= 0 && r_home >= 0 && g_want_rccl
                          && (sc->device != c->device || g_gather_forced);
    if (use_rccl) {
        // one group: the source rank sends on its stream (behind the pack), the home rank receives on the stack's stream
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.Send(s->d_pack_k, packed, ncclUint32, r_home, g_comms[r_src], s->st2);
        if (r == ncclSuccess) r = g_rccl.Recv(dst_k, packed, ncclUint32, r_src, g_comms[r_home], st->st);
        if (r == ncclSuccess) r = g_rccl.Send(s->d_pack_i, packed, ncclUint32, r_home, g_comms[r_src], s->st2);
        if (r == ncclSuccess) r = g_rccl.Recv(dst_i, packed, ncclUint32, r_src, g_comms[r_home], st->st);
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = re;
        if (r != ncclSuccess) {
            c->err = std::string("RCCL send/recv of the strong-pixel lists: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
            return FFS_ERR_DEVICE;
        }
        return FFS_OK;
    }
    // peer copy (or plain device-to-device when both contexts sit on one GPU), in the stack's stream, behind the pack
    STK_TRY(c, hipEventRecord(s->ev_pack, s->st2));
    HIP_TRY(c, hipSetDevice(c->device));
    STK_TRY(c, hipStreamWaitEvent(st->st, s->ev_pack, 0));
    if (sc->device == c->device) {
        STK_TRY(c, hipMemcpyAsync(dst_k, s->d_pack_k, packed * 4, hipMemcpyDeviceToDevice, st->st));
        STK_TRY(c, hipMemcpyAsync(dst_i, s->d_pack_i, packed * 4, hipMemcpyDeviceToDevice, st->st));
    } else {
        STK_TRY(c, hipMemcpyPeerAsync(dst_k, c->device, s->d_pack_k, sc->device, packed * 4, st->st));
        STK_TRY(c, hipMemcpyPeerAsync(dst_i, c->device, s->d_pack_i, sc->device, packed * 4, st->st));
    }
    if (s->ev_sent && s->ev_sent_dev != c->device) {   // (an event is recorded in a stream of its own device)
        (void)hipEventDestroy(s->ev_sent);
        s->ev_sent = nullptr;
    }
    if (!s->ev_sent) {
        STK_TRY(c, hipEventCreateWithFlags(&s->ev_sent, hipEventDisableTiming));
        s->ev_sent_dev = c->device;
    }
    STK_TRY(c, hipEventRecord(s->ev_sent, st->st));
    s->sent_pending = true;
    return FFS_OK;
}

// The lists of the stream's last batch go from its device buffers to the stack's, device to device.
static int stack3d_add_batch_impl(ffs_stack3d* st, ffs_stream* s) {
    ffs_ctx* c = s->ctx;
    if (s->busy || s->results.empty()) {
        c->err = "ffs_stack3d_add_batch: call after ffs_wait()";
        return FFS_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lock(st->mu);
    const uint32_t nf = s->n_frames;
    uint64_t more = 0;
    for (uint32_t f = 0; f < nf; ++f) more += s->results[f].num_strong_pixels;
    for (uint32_t f = 0; f < nf; ++f) {
        // a frame that did not fit the stream's lists was re-run on its one-frame stream: its list is on the host, if kept
        const OverflowFrame* o = overflow_of(s, f);
        if (o && s->results[f].num_strong_pixels && o->k.size() != s->results[f].num_strong_pixels) {
            c->err = "ffs_stack3d_add_batch: a frame overflowed the stream's lists; set want_strong_list (or a larger "
                     "max_strong_per_frame) for rotation sweeps";
            return FFS_ERR_OVERFLOW;
        }
    }
    ffs_ctx* home = st->ctx;
    const bool remote = s->ctx != st->ctx;
    if (remote && (home->L.W != c->L.W || home->L.H != c->L.H)) {
        c->err = "ffs_stack3d_add_batch: the stream's context has another frame shape than the stack's";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(home, hipSetDevice(home->device));
    int rc = stack3d_reserve(st, more);
    if (rc != FFS_OK) { if (remote) c->err = home->err; return rc; }
    // frame f's entries land at `at`; overflow frames go up from the host, the others from the stream's device lists
    uint64_t at = st->arrived;
    uint64_t packed = 0;
    uint32_t biggest = 0;
    bool host_lists = false;
    hipStream_t up_st = remote ? st->st : s->st2;   // (remote: the stack's stream on the home device)
    std::vector<uint64_t> dst(nf);
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t n = s->results[f].num_strong_pixels;
        dst[f] = at;
        at += n;
    }
    if (remote) {
        // packed frames must be contiguous in the stack: overflow frames' entries are placed behind them
        uint64_t a2 = st->arrived;
        for (uint32_t f = 0; f < nf; ++f)
            if (!overflow_of(s, f)) { dst[f] = a2; a2 += s->results[f].num_strong_pixels; }
        packed = a2 - st->arrived;
        for (uint32_t f = 0; f < nf; ++f)
            if (overflow_of(s, f)) { dst[f] = a2; a2 += s->results[f].num_strong_pixels; }
    }
    for (uint32_t f = 0; f < nf; ++f) {
        const OverflowFrame* o = overflow_of(s, f);
        const uint32_t n = s->results[f].num_strong_pixels;
        if (o && n) {
            STK_TRY(c, hipMemcpyAsync(st->a_k.p + dst[f], o->k.data(), (size_t)n * 4, hipMemcpyHostToDevice, up_st));
            STK_TRY(c, hipMemcpyAsync(st->a_i.p + dst[f], o->inten.data(), (size_t)n * 4, hipMemcpyHostToDevice, up_st));
            host_lists = true;
        }
        if (!o) biggest = std::max(biggest, n);
    }
    if (remote) {
        rc = stack3d_add_batch_remote(st, s, packed, biggest);
        if (rc != FFS_OK) { c->err = home->err; return rc; }
        if (host_lists) {   // (the overflow frames' lists live in the stream's result vectors)
            HIP_TRY(home, hipSetDevice(home->device));
            STK_TRY(c, hipStreamSynchronize(st->st));
        }
        for (uint32_t f = 0; f < nf; ++f)
            st->slices[s->results[f].frame_id] = ffs_stack3d::Slice{(uint32_t)dst[f], s->results[f].num_strong_pixels};
        st->arrived = at;
        (void)hipSetDevice(c->device);
        return FFS_OK;
    }
    // Local: the append runs in the stream's own sparse stream: behind the launch that wrote the lists, ahead of the one that
    // will overwrite them -- nothing to wait for here.  Its slice table sits in one of kSlots slots.
    const int slot = (int)(st->n_adds++ % ffs_stack3d::kSlots);
    if (st->ring_stride < c->max_batch || !st->d_ring.p) {
        stack3d_join_appends(st);
        st->ring_stride = std::max<size_t>(c->max_batch, nf);
        st->h_ring.pinned_host = true;
        STK_TRY(c, st->h_ring.ensure(st->ring_stride * ffs_stack3d::kSlots));
        STK_TRY(c, st->d_ring.ensure(st->ring_stride * ffs_stack3d::kSlots));
    }
    if (nf > st->ring_stride) {
        c->err = "ffs_stack3d_add_batch: batch larger than the context's max_batch";
        return FFS_ERR_INVALID;
    }
    if (!st->slot_ev[slot]) STK_TRY(c, hipEventCreateWithFlags(&st->slot_ev[slot], hipEventDisableTiming));
    if (st->slot_used[slot]) {
        STK_TRY(c, hipEventSynchronize(st->slot_ev[slot]));
        st->slot_used[slot] = false;
    }
    StackSlice* h_tab = st->h_ring.p + (size_t)slot * st->ring_stride;
    StackSlice* d_tab = st->d_ring.p + (size_t)slot * st->ring_stride;
    for (uint32_t f = 0; f < nf; ++f) {
        const ffs_frame_result& r = s->results[f];
        h_tab[f] = StackSlice{0u, (uint32_t)dst[f], overflow_of(s, f) ? 0u : r.num_strong_pixels, 0u};
        st->slices[r.frame_id] = ffs_stack3d::Slice{(uint32_t)dst[f], r.num_strong_pixels};
    }
    if (biggest) {
        STK_TRY(c, hipMemcpyAsync(d_tab, h_tab, (size_t)nf * sizeof(StackSlice), hipMemcpyHostToDevice, s->st2));
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_stack_append, dim3(std::min<uint32_t>(64, (biggest + 255) / 256), nf), dim3(256), 0, s->st2,
                           s->d_list_k, s->d_list_i, (uint64_t)s->cap, d_tab, st->a_k.p, st->a_i.p);
        STK_TRY(c, hipGetLastError());
    }
    STK_TRY(c, hipEventRecord(st->slot_ev[slot], s->st2));
    st->slot_used[slot] = true;
    if (host_lists) STK_TRY(c, hipStreamSynchronize(s->st2));  // (the overflow frames' lists live in the stream's result vectors)
    st->arrived = at;
    return FFS_OK;
}

extern "C" int ffs_stack3d_add_batch(ffs_stack3d* st, ffs_stream* s) {
    if (!st || !s) return FFS_ERR_INVALID;
    if (!s->lists_valid) {   // (the stack was created after the batch was submitted, or tuning "device_lists" is 0)
        s->ctx->err = "ffs_stack3d_add_batch: this batch did not leave its strong-pixel lists on the device -- create the stack before "
                      "submitting (tuning \"device_lists\" 2) or set \"device_lists\" to 1";
        return FFS_ERR_INVALID;
    }
    return guarded(s->ctx, [&] { return stack3d_add_batch_impl(st, s); });
}

extern "C" int ffs_stack3d_signals(ffs_stack3d* st, const uint32_t** x, const uint32_t** y, const int32_t** z,
                                   const uint32_t** intensity, const int32_t** reflection, uint64_t* n) {
    if (!st) return FFS_ERR_INVALID;
    if (x) *x = st->sig_x.data();
    if (y) *y = st->sig_y.data();
    if (z) *z = st->sig_z.data();
    if (intensity) *intensity = st->sig_i.data();
    if (reflection) *reflection = st->sig_refl.data();
    if (n) *n = st->sig_refl.size();
    return FFS_OK;
}

static int stack3d_finish_impl(ffs_stack3d* st, const ffs_reflection** reflections, uint32_t* n_refl,
                               uint32_t* n_calculated, uint32_t* n_f_size, uint32_t* n_f_sep) {
    ffs_ctx* c = st->ctx;
    std::lock_guard<std::mutex> lock(st->mu);
    HIP_TRY(c, hipSetDevice(c->device));
    for (int k = 0; k < ffs_stack3d::kSlots; ++k)   // the appends of the last batches may still be running (in other streams)
        if (st->slot_used[k]) HIP_TRY(c, hipStreamWaitEvent(st->st, st->slot_ev[k], 0));
    // z = rank of the frame id among the slices held (std::map order, spotfinder.cc:1105-1108)
    const int nz = (int)st->slices.size();
    uint64_t total = 0;
    for (auto& kv : st->slices) total += kv.second.n;
    st->out.clear();
    st->sig_x.clear(); st->sig_y.clear(); st->sig_z.clear(); st->sig_i.clear(); st->sig_refl.clear();
    uint32_t n_calc = 0, fs = 0, fp = 0;
    if (total > 0) {
        const uint32_t N = (uint32_t)total;
        const uint32_t chunks = (N + kRootChunk - 1) / kRootChunk;
        STK_TRY(c, st->h_table.ensure(nz));
        STK_TRY(c, st->d_table.ensure(nz));
        STK_TRY(c, st->d_begin.ensure((size_t)nz + 1));
        STK_TRY(c, st->d_k.ensure(N)); STK_TRY(c, st->d_i.ensure(N)); STK_TRY(c, st->d_z.ensure(N));
        STK_TRY(c, st->d_parent.ensure(N)); STK_TRY(c, st->d_comp.ensure(N));
        STK_TRY(c, st->d_acc.ensure(N)); STK_TRY(c, st->d_recs.ensure(N));
        STK_TRY(c, st->d_chunk_roots.ensure(chunks));
        STK_TRY(c, st->d_small.ensure(16));  // [0] n, [1] n_comp, [2] status, [8..15] summary
        STK_TRY(c, st->d_sx.ensure(N)); STK_TRY(c, st->d_sy.ensure(N)); STK_TRY(c, st->d_sc.ensure(N));
        std::vector<uint32_t> begin(nz + 1, 0);
        {
            int z = 0;
            uint32_t at = 0, biggest = 0;
            for (auto& kv : st->slices) {
                begin[z] = at;
                st->h_table.p[z] = StackSlice{kv.second.off, at, kv.second.n, (uint32_t)z};
                biggest = std::max(biggest, kv.second.n);
                at += kv.second.n;
                ++z;
            }
            begin[nz] = at;
            hipEvent_t e0, e1;
            STK_TRY(c, hipEventCreate(&e0));
            STK_TRY(c, hipEventCreate(&e1));
            const uint32_t small[16] = {N, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipEventRecord(e0, st->st);
            STK_TRY(c, hipMemcpyAsync(st->d_table.p, st->h_table.p, (size_t)nz * sizeof(StackSlice), hipMemcpyHostToDevice, st->st));
            STK_TRY(c, hipMemcpyAsync(st->d_begin.p, begin.data(), (size_t)(nz + 1) * 4, hipMemcpyHostToDevice, st->st));
            STK_TRY(c, hipMemcpyAsync(st->d_small.p, small, sizeof(small), hipMemcpyHostToDevice, st->st));
            SegArgs sa{};
            sa.list_k = st->d_k.p;
            sa.list_i = st->d_i.p;
            sa.parent = st->d_parent.p;
            sa.comp_id = st->d_comp.p;
            sa.seg_n = st->d_small.p;
            sa.seg_stride = N;
            sa.n_comp = st->d_small.p + 1;
            sa.acc = st->d_acc.p;
            sa.max_comp = N;
            sa.overflow = st->d_small.p + 2;
            sa.W = (uint32_t)c->L.W;
            sa.H = (uint32_t)c->L.H;
            sa.row_off = nullptr;
            sa.slice_begin = st->d_begin.p;
            sa.n_slices = nz;
            sa.zs = st->d_z.p;
            sa.min_spot_size = c->params.min_spot_size_3d;
            sa.max_sep = c->params.max_peak_centroid_separation;
            sa.recs = st->d_recs.p;
            sa.summary = st->d_small.p + 8;
            sa.chunk_roots = st->d_chunk_roots.p;
            sa.chunks_max = chunks;
            (void)hipGetLastError();
            const unsigned nb = (unsigned)std::min<uint64_t>(4096, ((uint64_t)N + 255) / 256);
            hipLaunchKernelGGL(k_stack_gather, dim3(std::min<uint32_t>(64, (biggest + 255) / 256), nz), dim3(256), 0, st->st,
                               st->a_k.p, st->a_i.p, st->d_table.p, st->d_k.p, st->d_i.p, st->d_z.p, st->d_parent.p, st->d_acc.p);
            hipLaunchKernelGGL(k_union<true>, dim3(nb, 1), dim3(256), 0, st->st, sa);
            hipLaunchKernelGGL(k_reduce_roots3d, dim3(std::min<unsigned>(chunks, 2048u)), dim3(256), 0, st->st, sa);
            hipLaunchKernelGGL(k_finalize_roots3d, dim3(std::min<unsigned>(chunks, 1024u)), dim3(256), 0, st->st, sa);
            hipLaunchKernelGGL(k_stack_labels, dim3(nb), dim3(256), 0, st->st, sa, st->d_sx.p, st->d_sy.p, st->d_sc.p);
            STK_TRY(c, hipGetLastError());
            uint32_t back[16];
            STK_TRY(c, hipMemcpyAsync(back, st->d_small.p, sizeof(back), hipMemcpyDeviceToHost, st->st));
            (void)hipEventRecord(e1, st->st);
            STK_TRY(c, hipStreamSynchronize(st->st));
            (void)hipEventElapsedTime(&st->last_finish_ms, e0, e1);
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            n_calc = back[1];
        }
        std::vector<ReflOut> recs(n_calc);
        if (n_calc) STK_TRY(c, hipMemcpyAsync(recs.data(), st->d_recs.p, (size_t)n_calc * sizeof(ReflOut), hipMemcpyDeviceToHost, st->st));
        std::vector<uint32_t> comp(N), zs(N);
        st->sig_x.resize(N); st->sig_y.resize(N); st->sig_i.resize(N);
        STK_TRY(c, hipMemcpyAsync(st->sig_x.data(), st->d_sx.p, (size_t)N * 4, hipMemcpyDeviceToHost, st->st));
        STK_TRY(c, hipMemcpyAsync(st->sig_y.data(), st->d_sy.p, (size_t)N * 4, hipMemcpyDeviceToHost, st->st));
        STK_TRY(c, hipMemcpyAsync(st->sig_i.data(), st->d_i.p, (size_t)N * 4, hipMemcpyDeviceToHost, st->st));
        STK_TRY(c, hipMemcpyAsync(comp.data(), st->d_sc.p, (size_t)N * 4, hipMemcpyDeviceToHost, st->st));
        STK_TRY(c, hipMemcpyAsync(zs.data(), st->d_z.p, (size_t)N * 4, hipMemcpyDeviceToHost, st->st));
        STK_TRY(c, hipStreamSynchronize(st->st));
        std::vector<int32_t> kept_index(n_calc, -1);
        for (uint32_t q = 0; q < n_calc; ++q) {
            const ReflOut& r = recs[q];
            if (r.flags & 1u) ++fs;
            else if (r.flags & 2u) ++fp;
            else {
                ffs_reflection o;
                std::memcpy(&o, &r, sizeof(o));
                kept_index[q] = (int32_t)st->out.size();
                st->out.push_back(o);
            }
        }
        st->sig_z.resize(N);
        st->sig_refl.resize(N);
        for (uint32_t i = 0; i < N; ++i) {
            st->sig_z[i] = (int32_t)zs[i];
            st->sig_refl[i] = comp[i] < n_calc ? kept_index[comp[i]] : -1;
        }
    }
    if (reflections) *reflections = st->out.data();
    if (n_refl) *n_refl = (uint32_t)st->out.size();
    if (n_calculated) *n_calculated = n_calc;
    if (n_f_size) *n_f_size = fs;
    if (n_f_sep) *n_f_sep = fp;
    return FFS_OK;
}

extern "C" int ffs_stack3d_finish(ffs_stack3d* st, const ffs_reflection** reflections, uint32_t* n_refl,
                                  uint32_t* n_calculated, uint32_t* n_f_size, uint32_t* n_f_sep) {
    if (!st) return FFS_ERR_INVALID;
    return guarded(st->ctx, [&] { return stack3d_finish_impl(st, reflections, n_refl, n_calculated, n_f_size, n_f_sep); });
}
extern "C" int ffs_stack3d_last_finish_ms(const ffs_stack3d* st, float* ms) {
    if (!st || !ms) return FFS_ERR_INVALID;
    *ms = st->last_finish_ms;
    return FFS_OK;
}
