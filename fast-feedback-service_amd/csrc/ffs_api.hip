// ffs_api.hip -- host side of libffs_hip.so: contexts, streams, launches, result assembly.
// The C ABI is declared in include/ffs_hip.h; every entry point there names the reference
// interface it replaces.  No exceptions leave this file.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is loaded with dlopen when several devices are in use

#include "ffs_hip.h"
#include "kernels_ccl.hpp"
#include "kernels_threshold.hpp"
#include "kernels_extended.hpp"
#include "kernels_stream.hpp"
#include "kernels_stack3d.hpp"
#include "kernels_chain.hpp"
#include "kernels_decode.hpp"

using namespace ffsamd;

static_assert(sizeof(ReflOut) == sizeof(ffs_reflection), "record layout");
static_assert(offsetof(ReflOut, sum_intensity) == offsetof(ffs_reflection, sum_intensity), "record layout");

static thread_local std::string g_create_error;

// Error text is kept per calling thread (several worker threads drive their own streams of one
// context; a shared std::string would be a data race exactly when things go wrong).  `ctx->err = ...`
// and `ctx->err.c_str()` keep reading naturally at the call sites.
struct ThreadError {
    static std::string& text() {
        static thread_local std::string t;
        return t;
    }
    const ThreadError& operator=(const std::string& m) const { text() = m; return *this; }
    const ThreadError& operator=(const char* m) const { text() = m; return *this; }
    const char* c_str() const { return text().c_str(); }
    operator std::string() const { return text(); }
};

// Environment knobs (A/B switches of the measurements in DESIGN.md) are read ONCE, when a context is created:
// nothing on the submit path calls getenv (which is not safe against a setenv in another thread of the host).
struct Knobs {
    int k1_variant = 2, ext_variant = 2, k1_debug = 0, k1_group = 1 << 30, k1_ahead = 2;
    long long target_waves = 16384;
    int emit_variant = 1, ccl_variant = 2, link_runs = 1, ccl_grid = 32, ccl_cus = 0, direct_recs = 1;
    int sched = 3;             // FFS_SCHED: 1 = sparse chain on a high-priority stream of its own; 2 = also every dense kernel of the context on ONE stream
    int chain_stop = 0, dummy_us = 0, dummy_wg = 32, dummy_threads = 1024, dummy_lds = 0;   // FFS_CHAIN_STOP, FFS_DUMMY_*: timing experiments
    int chain_first = 2;       // FFS_CHAIN_FIRST = n > 0: the sparse launch also does the bright-window fix-up, and while at most n batches are in flight a streaming kernel waits until the previous batch's sparse launch has started
    int ext_launch = 1;        // FFS_EXT_LAUNCH: the streaming kernel carries its start / stop events (hipExtLaunchKernel)
    int use_occ = 1;           // FFS_OCC: k_frame_chain reads only the plane segments the occupancy bitmap names
    int fix_aside = 1;         // FFS_FIX_ASIDE: k_bright_fix in the sparse stream (SCHED >= 1)
    int decode_dense = 1;      // FFS_DECODE_DENSE: the decode kernel runs in the dense kernels' stream (0: in the upload stream)
    int dense_mask = 0;        // FFS_DENSE_MASK=1: always produce the dense byte mask
    int chain_skip = 0;        // FFS_CHAIN_SKIP (timing experiments only; results are then meaningless): 1 = no sparse chain, 2 = stop after emit, 4 = after union, 8 = after reduce
    int bright_cap = 1 << 20;  // FFS_BRIGHT_CAP: entries of the bright-window list actually used (tests shrink it)
    static int env_int(const char* name, int dflt) {
        const char* e = std::getenv(name);
        return e ? std::atoi(e) : dflt;
    }
    void read() {
        k1_variant = env_int("FFS_K1_VARIANT", 2);
        ext_variant = env_int("FFS_EXT_VARIANT", 2);
        k1_debug = env_int("FFS_K1_DEBUG", 0);
        k1_group = std::max(1, env_int("FFS_K1_GROUP", 1 << 30));
        k1_ahead = env_int("FFS_K1_AHEAD", 2);
        target_waves = std::max(1, env_int("FFS_K1_TARGET_WAVES", 16384));
        emit_variant = env_int("FFS_EMIT", 1);
        ccl_variant = env_int("FFS_CCL", 2);
        link_runs = env_int("FFS_LINK_RUNS", 1);
        ccl_grid = std::max(1, env_int("FFS_CCL_GRID", 32));
        ccl_cus = env_int("FFS_CCL_CUS", 0);
        direct_recs = env_int("FFS_DIRECT_RECS", 1);
        chain_skip = env_int("FFS_CHAIN_SKIP", 0);
        sched = env_int("FFS_SCHED", 3);
        dense_mask = env_int("FFS_DENSE_MASK", 0);
        decode_dense = env_int("FFS_DECODE_DENSE", 1);
        fix_aside = env_int("FFS_FIX_ASIDE", 1);
        use_occ = env_int("FFS_OCC", 1);
        ext_launch = env_int("FFS_EXT_LAUNCH", 1);
        chain_first = env_int("FFS_CHAIN_FIRST", 2);
        chain_stop = env_int("FFS_CHAIN_STOP", 0);
        dummy_us = env_int("FFS_DUMMY_US", 0);
        dummy_wg = std::max(1, env_int("FFS_DUMMY_WG", 32));
        dummy_threads = std::max(64, std::min(1024, env_int("FFS_DUMMY_THREADS", 1024)));
        dummy_lds = std::max(0, std::min(65536, env_int("FFS_DUMMY_LDS", 0)));
        bright_cap = std::max(0, std::min(1 << 20, env_int("FFS_BRIGHT_CAP", 1 << 20)));
    }
};

struct ffs_ctx {
    int device = 0;
    Knobs knobs;
    Layout L{};
    int pixel_bytes = 2;
    uint32_t max_batch = 1;
    uint32_t cap = 0;       // strong pixels per frame
    uint32_t max_comp = 0;  // components per frame
    int n_tiles = 0, n_strips = 0;
    ffs_params params{};
    uint8_t* d_maskbits = nullptr;
    uint8_t* d_ginfo = nullptr;  // per-group mask bits + window-count bounds (kernels_stream.hpp)
    uint8_t* d_mmap = nullptr;   // per-pixel window counts
    hipStream_t dense_st = nullptr;  // FFS_SCHED=2: the one stream of the dense kernels
    hipStream_t dense_st_b = nullptr;
    hipStream_t up_st = nullptr;     // ... and the one stream of uploads and decoding
    hipStream_t sparse_st[2] = {nullptr, nullptr};  // FFS_SCHED=3: the sparse chains of the context's streams, alternating
    int n_streams_made = 0;
    std::mutex stream_mu;            // guards the lazy creation of the shared streams
    std::vector<struct ffs_stack3d*> stack_pool;   // destroyed 3D stacks kept with their buffers for the next sweep (stream_mu)
    std::atomic<int> inflight{0};    // batches between submit and wait, over all ffs_streams of the context
    std::atomic<hipEvent_t> last_chain_start{nullptr};   // start event of the newest sparse launch: the next streaming kernel lets it get its CUs first
    bool chain_ok = false;           // k_frame_chain may use its 140 KB of dynamic LDS on this device
    ThreadError err;  // the calling thread's most recent error on any context
};

// ---- no exception crosses the C ABI -----------------------------------------------------------------------
// The entry points that grow std::vectors (results, staging tables, masks) run inside a catch-all: an
// allocation failure or a length error becomes FFS_ERR_NOMEM with its text in ffs_last_error, instead of
// std::terminate -> abort() in the caller's process.
template <typename F>
static int guarded(ffs_ctx* c, F&& body) {
    try {
        return body();
    } catch (const std::exception& e) {
        if (c) c->err = std::string("exception inside libffs_hip: ") + e.what();
        else g_create_error = std::string("exception inside libffs_hip: ") + e.what();
        return FFS_ERR_NOMEM;
    } catch (...) {
        if (c) c->err = "unknown exception inside libffs_hip";
        return FFS_ERR_NOMEM;
    }
}


#define HIP_TRY(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);             \
            return e_ == hipErrorOutOfMemory ? FFS_ERR_NOMEM : FFS_ERR_DEVICE;          \
        }                                                                               \
    } while (0)

// per-tile strong counts of a batch + one word (the last) for the bright-list count; a multiple of 256 bytes so
// that one fill kernel clears it
struct ffs_stream;
static size_t tile_counts_bytes(const ffs_stream* s);

constexpr uint32_t kBrightCap = 1u << 20;  // entries of the bright-window list per batch (8 MB)

struct OverflowFrame;

struct ffs_stream {
    ffs_ctx* ctx = nullptr;
    uint32_t max_batch = 1;   // frames per submit
    uint32_t cap = 0;         // strong pixels per frame the lists hold
    uint32_t max_comp = 0;    // components per frame the record buffers hold
    ffs_stream* big = nullptr;               // one-frame stream with room for frames that exceed cap / max_comp
    std::vector<OverflowFrame> ovf;          // such frames of the last batch, re-run on `big`
    int force_variant = -1;                  // >= 0: threshold variant of the next enqueue (bright-list overflow -> 1)
    uint32_t *d_pack_k = nullptr, *d_pack_i = nullptr;  // a batch's lists packed end to end for another device's 3D stack
    StackSlice *d_pack_tab = nullptr, *h_pack_tab = nullptr;
    hipStream_t st = nullptr;    // threshold kernels (+ H2D)
    hipStream_t st_up = nullptr; // uploads + decode; == st unless the dense kernels of the context share one stream
    bool st2_shared = false;
    bool st_shared = false;      // st is the context's dense stream (not ours to destroy)
    hipStream_t st2 = nullptr;   // compaction + connected components + D2H; == st unless the CUs are split
    hipEvent_t ev[7] = {};   // [6]: the compressed chunks and their block table are on the device
    // device
    uint8_t* d_img = nullptr;
    uint8_t* d_bits = nullptr;
    uint8_t* d_sbytes = nullptr;
    uint8_t *d_dplane = nullptr, *d_eplane = nullptr;  // extended algorithm only (allocated on first use)
    uint8_t* d_comp = nullptr;                         // compressed chunks (allocated on first use)
    uint2 *d_tab = nullptr, *h_tab = nullptr;          // per-block (offset, length) tables
    uint32_t dec_blocks = 0, dec_last = 0, dec_tail = 0, dec_block_elems = 0;
    // Records are copied back speculatively with the counts (one wait instead of two): room for the most
    // records per frame seen so far on this stream, +25 %; ffs_wait fetches the rest if a batch exceeds it.
    uint32_t spec_recs_per_frame = 256;
    uint64_t spec_recs_copied = 0;
    std::thread job;          // ffs_submit_compressed's helper (block index + launches); joined by ffs_wait
    int job_rc = 0;
    std::string job_err;
    uint32_t *d_tile_counts = nullptr, *d_num_strong = nullptr, *d_row_off = nullptr;
    uint2* d_bright = nullptr;  // pixels k_stream_u16 hands to k_bright_fix; their count sits behind the tile counts
    uint32_t *d_list_k = nullptr, *d_list_i = nullptr, *d_parent = nullptr, *d_comp_id = nullptr;
    uint32_t *d_n_comp = nullptr, *d_overflow = nullptr, *d_summary = nullptr;
    uint32_t* d_part_roots = nullptr;
    CompAcc* d_acc = nullptr;
    CompAcc2* d_acc2 = nullptr;          // accumulators at the root's list index (2D, k_reduce_roots)
    uint32_t* d_chunk_roots = nullptr;
    bool wire2 = false;                  // the batch in flight ships 40-byte WireRec2 records
    ReflOut* d_recs = nullptr;
    // pinned host
    uint8_t* h_img = nullptr;
    size_t h_img_bytes = 0;
    uint32_t* h_counts = nullptr;  // [max_batch] num_strong | [max_batch] n_comp | [max_batch*8] summary | [1] overflow
    ReflOut* h_recs = nullptr;
    uint32_t* d_occ = nullptr;     // [max_batch][occ_frame_words] occupancy of the strong plane (one bit per 16-byte segment)
    uint32_t* h_counts_dev = nullptr;  // device-side address of h_counts (k_frame_chain writes the counters itself)
    hipEvent_t ev_cs = nullptr;    // start of this stream's k_frame_chain launch (rides on the dispatch)
    bool ev1_pending = false;      // ev[1] (start of the threshold stage) has not been recorded yet for this batch
    bool ev3_is_ev4 = false;       // one event behind the sparse launch (k_frame_chain leaves nothing to copy)
    bool dev_input = false;        // this batch's frames were on the device already (ffs_submit_device): no upload, no ev[0]
    bool dense_valid = false;      // the byte masks of the last batch were produced
    bool chain_mode = false;       // this batch went through k_frame_chain: records at frame * max_comp, flags per frame
    ReflOut* h_recs_dev = nullptr;  // device-side address of h_recs when the records are written straight to the host
    bool direct_recs = false;
    bool bits_cleared = false;  // the last batch's compaction zeroed the strong plane again (k_stream_u16's invariant)
    bool bits_dirty = false;    // the strong plane may hold bits: k_stream_u16 needs it zeroed first
    bool counts_dirty = true;   // the per-tile counts (+ bright-list count) may be non-zero: the streaming kernels add into them
    uint32_t *h_list_k = nullptr, *h_list_i = nullptr;
    uint8_t* h_mask = nullptr;
    // state of the batch in flight
    bool busy = false;
    uint32_t n_frames = 0;
    int64_t first_id = 0;
    const void* cur_img = nullptr;
    size_t cur_pitch = 0, cur_fstride = 0;
    ffs_params batch_params{};
    float timings[5] = {0, 0, 0, 0, 0};
    // results
    std::vector<ffs_frame_result> results;
    std::vector<ffs_box> boxes;
    std::vector<ffs_reflection> refls;
};

// Results of a frame that did not fit the stream's lists, from its re-run on the one-frame stream
struct OverflowFrame {
    uint32_t frame = 0;
    ffs_frame_result res{};
    std::vector<ffs_box> boxes;
    std::vector<ffs_reflection> refls;
    std::vector<uint32_t> k, inten;
};

static inline void mark_busy(ffs_stream* s);
static inline void mark_idle(ffs_stream* s);

static uint32_t occ_frame_words(const Layout& L) { return (uint32_t)(((uint64_t)L.H * (L.mpitch / 16) + 31) / 32 + 2); }  // (+2: the chain reads a word ahead)

// per-tile counts | ... | [last - 1] workgroups of k_frame_chain through with the bright list | [last] entries of the bright list
static inline void mark_busy(ffs_stream* s) {
    if (!s->busy) ++s->ctx->inflight;
    s->busy = true;
}
static inline void mark_idle(ffs_stream* s) {
    if (s->busy) --s->ctx->inflight;
    s->busy = false;
}
static size_t tile_counts_bytes(const ffs_stream* s) { return (((size_t)s->max_batch * s->ctx->n_tiles + 2) * 4 + 255) / 256 * 256; }

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

// ---------------------------------------------------------------------------------------------------

extern "C" void ffs_default_params(ffs_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->min_count = 2;  // baseline/spotfinder/standalone.cc:17
    p->nsig_b = 6.0;   // :19
    p->nsig_s = 3.0;   // :20
    p->threshold = 0.0;
    p->max_valid = -1;
    p->min_spot_size = 3;     // spotfinder/spotfinder.cc:321
    p->min_spot_size_3d = 3;  // :327
    p->max_peak_centroid_separation = 2.0f;  // :335
    p->want_reflections = 1;
    p->want_strong_list = 0;
    p->want_strong_mask = 0;
    p->algorithm = FFS_ALGO_DISPERSION;
    p->extended_flavour = 0;
}

extern "C" int ffs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ffs_device_name(int device, char* buf, size_t buflen) {
    hipDeviceProp_t prop;
    if (!buf || buflen == 0) return FFS_ERR_INVALID;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FFS_ERR_NODEVICE;
    std::snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return FFS_OK;
}

extern "C" int ffs_device_total_mem(int device, uint64_t* bytes) {
    hipDeviceProp_t prop;
    if (!bytes) return FFS_ERR_INVALID;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FFS_ERR_NODEVICE;
    *bytes = prop.totalGlobalMem;
    return FFS_OK;
}

extern "C" const char* ffs_last_error(const ffs_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();  // both are per-thread texts
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int ffs_ctx_create(int device, uint32_t width, uint32_t height, int pixel_bytes,
                              uint32_t max_batch, uint32_t max_strong, ffs_ctx** out) {
    if (!out) return FFS_ERR_INVALID;
    *out = nullptr;
    if (width == 0 || height == 0 || (pixel_bytes != 2 && pixel_bytes != 4) || max_batch == 0
        || width > 10240 || (uint64_t)width * height >= (1ull << 32)) {
        g_create_error = "ffs_ctx_create: need 0 < width <= 10240, width*height < 2^32, pixel_bytes 2 or 4, max_batch > 0";
        return FFS_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_create_error = "no HIP device visible (libffs_hip.so has no CPU fallback)";
        return FFS_ERR_NODEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "ffs_ctx_create: device index out of range";
        return FFS_ERR_NODEVICE;
    }
    ffs_ctx* c = new (std::nothrow) ffs_ctx();
    if (!c) return FFS_ERR_NOMEM;
    c->device = device;
    c->knobs.read();
    c->pixel_bytes = pixel_bytes;
    c->max_batch = max_batch;
    Layout& L = c->L;
    L.W = (int)width;
    L.H = (int)height;
    L.pitch_px = round_up((int)width, 128);  // byte-mask rows start on 128-byte lines
    L.pitch = (uint32_t)L.pitch_px * (uint32_t)pixel_bytes;
    L.mpitch = (uint32_t)L.pitch_px / 8;
    L.bpitch = (uint32_t)L.pitch_px;
    L.frame_stride = (uint64_t)L.pitch * height;
    L.plane_frame_stride = (uint64_t)L.mpitch * height;
    L.bytes_frame_stride = (uint64_t)L.bpitch * height;
    if (L.frame_stride >= (1ull << 32)) {
        g_create_error = "frame larger than 4 GiB";
        delete c;
        return FFS_ERR_INVALID;
    }
    const uint64_t npx = (uint64_t)width * height;
    c->cap = max_strong ? max_strong : (uint32_t)std::min<uint64_t>(npx, 1u << 18);
    c->cap = (uint32_t)std::min<uint64_t>(c->cap, npx);
    c->max_comp = std::min<uint32_t>(c->cap, 1u << 16);
    c->n_tiles = ((int)height + kTileRows - 1) / kTileRows;
    const int owned_px = pixel_bytes == 2 ? kStripOwnedPx : kStripOwnedPx32;
    c->n_strips = (L.pitch_px + owned_px - 1) / owned_px;
    ffs_default_params(&c->params);
    if (hipSetDevice(device) != hipSuccess) {
        g_create_error = "hipSetDevice failed";
        delete c;
        return FFS_ERR_DEVICE;
    }
    hipError_t e = hipMalloc(&c->d_maskbits, L.plane_frame_stride + 256);
    // one dword per lane group of 16 bytes of pixels: 8 pixels (16-bit) or 4 pixels (32-bit)
    if (e == hipSuccess) e = hipMalloc(&c->d_ginfo, (size_t)(L.H + kInfoExtraRows) * ((size_t)L.pitch_px * pixel_bytes / 4) + 256);
    if (e == hipSuccess) e = hipMalloc(&c->d_mmap, (size_t)L.H * L.pitch_px + 256);
    if (e != hipSuccess) {
        g_create_error = std::string("hipMalloc(mask): ") + hipGetErrorString(e);
        ffs_ctx_destroy(c);
        return FFS_ERR_NOMEM;
    }
    *out = c;
    int rc = ffs_ctx_set_mask(c, nullptr);
    if (rc != FFS_OK) {
        g_create_error = c->err;
        ffs_ctx_destroy(c);
        *out = nullptr;
        return rc;
    }
    return FFS_OK;
}

static void stack3d_free(struct ffs_stack3d* st);

extern "C" void ffs_ctx_destroy(ffs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto* st : c->stack_pool) stack3d_free(st);
    c->stack_pool.clear();
    if (c->d_maskbits) (void)hipFree(c->d_maskbits);
    if (c->d_ginfo) (void)hipFree(c->d_ginfo);
    if (c->d_mmap) (void)hipFree(c->d_mmap);
    if (c->dense_st) (void)hipStreamDestroy(c->dense_st);
    if (c->up_st) (void)hipStreamDestroy(c->up_st);
    if (c->dense_st_b) (void)hipStreamDestroy(c->dense_st_b);
    for (auto st : c->sparse_st) if (st) (void)hipStreamDestroy(st);
    delete c;
}

// The tables of the one-kernel threshold path depend on the mask alone: rebuilt whenever it changes.
static int rebuild_mask_tables(ffs_ctx* c) {
    const Layout& L = c->L;
    const uint32_t gpitch = (uint32_t)L.pitch_px * (uint32_t)c->pixel_bytes / 4;
    HIP_TRY(c, hipMemset(c->d_ginfo, 0, (size_t)(L.H + kInfoExtraRows) * gpitch));
    const int groups = L.pitch_px / (c->pixel_bytes == 2 ? 8 : 4);
    (void)hipGetLastError();  // drop any stale error state: the check below is for this launch
    if (c->pixel_bytes == 2)
        hipLaunchKernelGGL(k_build_maps, dim3((groups + 255) / 256, L.H), dim3(256), 0, 0, c->d_maskbits, L.mpitch, L.W, L.H,
                           L.pitch_px, c->d_mmap, c->d_ginfo, gpitch);
    else
        hipLaunchKernelGGL(k_build_maps4, dim3((groups + 255) / 256, L.H), dim3(256), 0, 0, c->d_maskbits, L.mpitch, L.W, L.H,
                           L.pitch_px, c->d_mmap, c->d_ginfo, gpitch);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    return FFS_OK;
}

static int ffs_ctx_set_mask_impl(ffs_ctx* c, const uint8_t* host_mask) {
    if (!c) return FFS_ERR_INVALID;
    const Layout& L = c->L;
    std::vector<uint8_t> bits(L.plane_frame_stride, 0);
    for (int y = 0; y < L.H; ++y) {
        uint8_t* row = bits.data() + (size_t)y * L.mpitch;
        if (host_mask) {
            const uint8_t* m = host_mask + (size_t)y * L.W;
            for (int x = 0; x < L.W; ++x)
                if (m[x]) row[x >> 3] |= (uint8_t)(1u << (x & 7));
        } else {
            for (int x = 0; x < L.W; ++x) row[x >> 3] |= (uint8_t)(1u << (x & 7));
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(c->d_maskbits, bits.data(), bits.size(), hipMemcpyHostToDevice));
    return rebuild_mask_tables(c);
}

static int ffs_ctx_get_mask_impl(ffs_ctx* c, uint8_t* host_mask) {
    if (!c || !host_mask) return FFS_ERR_INVALID;
    const Layout& L = c->L;
    std::vector<uint8_t> bits(L.plane_frame_stride);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(bits.data(), c->d_maskbits, bits.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.H; ++y)
        for (int x = 0; x < L.W; ++x)
            host_mask[(size_t)y * L.W + x] = (bits[(size_t)y * L.mpitch + (x >> 3)] >> (x & 7)) & 1;
    return FFS_OK;
}

// Resolution mask: spotfinder/kernels/masking.cu:37-73 (float32 distance / d-spacing), :99-147.
// One thread per mask byte (8 pixels); only clears bits, as the reference only ever masks.
__global__ void k_resolution_mask(uint8_t* maskbits, uint32_t mpitch, int W, int H, float wavelength,
                                  float distance, float cx, float cy, float psx, float psy,
                                  float dmin, float dmax) {
    const int bx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (bx * 8 >= W || y >= H) return;
    uint8_t* p = maskbits + (uint64_t)y * mpitch + bx;
    uint32_t b = *p;
    for (int j = 0; j < 8; ++j) {
        const int x = bx * 8 + j;
        if (x >= W || !((b >> j) & 1u)) continue;  // masking.cu:120-126
        const float dx = (((float)x + 0.5f) - cx) * psx;  // :50-52
        const float dy = (((float)y + 0.5f) - cy) * psy;
        const float r = sqrtf(dx * dx + dy * dy);
        const float theta = 0.5f * atanf(r / distance);  // :71
        const float res = wavelength / (2.0f * sinf(theta));  // :72
        if ((dmin > 0 && res < dmin) || (dmax > 0 && res > dmax)) b &= ~(1u << j);  // :133-142
    }
    *p = (uint8_t)b;
}

extern "C" int ffs_ctx_apply_resolution_mask(ffs_ctx* c, float wavelength, float distance_m,
                                             float bcx, float bcy, float psx, float psy, float dmin,
                                             float dmax) {
    if (!c) return FFS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const Layout& L = c->L;
    dim3 block(64, 1), grid((L.mpitch + 63) / 64, L.H);
    hipLaunchKernelGGL(k_resolution_mask, grid, block, 0, 0, c->d_maskbits, L.mpitch, L.W, L.H,
                       wavelength, distance_m, bcx, bcy, psx, psy, dmin, dmax);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    return rebuild_mask_tables(c);
}

extern "C" int ffs_ctx_set_params(ffs_ctx* c, const ffs_params* p) {
    if (!c || !p) return FFS_ERR_INVALID;
    if (p->min_count < 2 || p->min_count > 49 || p->nsig_b < 0 || p->nsig_s < 0 || p->threshold < 0) {
        c->err = "ffs_ctx_set_params: need 2 <= min_count <= 49, nsig_b >= 0, nsig_s >= 0, threshold >= 0";
        return FFS_ERR_INVALID;  // the asserts of standalone.cc:52-63
    }
    if ((p->algorithm != FFS_ALGO_DISPERSION && p->algorithm != FFS_ALGO_DISPERSION_EXTENDED)
        || (p->extended_flavour != 0 && p->extended_flavour != 1)) {
        c->err = "ffs_ctx_set_params: unknown algorithm / extended_flavour";
        return FFS_ERR_INVALID;
    }
    c->params = *p;
    return FFS_OK;
}

extern "C" int ffs_ctx_device_layout(const ffs_ctx* c, size_t* pitch, size_t* fstride) {
    if (!c) return FFS_ERR_INVALID;
    if (pitch) *pitch = c->L.pitch;
    if (fstride) *fstride = c->L.frame_stride;
    return FFS_OK;
}

// ---- streams ---------------------------------------------------------------------------------------

template <typename T>
static hipError_t dmalloc(T** p, size_t n_bytes) {
    return hipMalloc(reinterpret_cast<void**>(p), n_bytes + 256);
}

extern "C" void ffs_stream_destroy(ffs_stream* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    if (s->job.joinable()) s->job.join();
    mark_idle(s);   // (a stream may be closed with its batch still in flight)
    if (s->big) ffs_stream_destroy(s->big);
    if (s->st_up && s->st_up != s->st) (void)hipStreamSynchronize(s->st_up);
    if (s->st) (void)hipStreamSynchronize(s->st);
    if (s->st2 && s->st2 != s->st) { (void)hipStreamSynchronize(s->st2); if (!s->st2_shared) (void)hipStreamDestroy(s->st2); }
    // (d_n_comp, d_summary and d_overflow live inside the d_num_strong allocation)
    if (s->h_pack_tab) (void)hipHostFree(s->h_pack_tab);
    void* dev[] = {s->d_occ, s->d_pack_k, s->d_pack_i, s->d_pack_tab, s->d_acc2, s->d_chunk_roots, s->d_bright, s->d_comp, s->d_tab, s->d_dplane, s->d_eplane, s->d_row_off, s->d_img, s->d_bits, s->d_sbytes, s->d_tile_counts, s->d_num_strong,
                   s->d_list_k, s->d_list_i, s->d_parent, s->d_comp_id, s->d_part_roots, s->d_acc, s->d_recs};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    void* host[] = {s->h_tab, s->h_img, s->h_counts, s->h_recs, s->h_list_k, s->h_list_i, s->h_mask};
    for (void* p : host)
        if (p) (void)hipHostFree(p);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->ev_cs) {
        hipEvent_t mine = s->ev_cs;
        s->ctx->last_chain_start.compare_exchange_strong(mine, nullptr);
        (void)hipEventDestroy(s->ev_cs);
    }

    if (s->st && !s->st_shared) (void)hipStreamDestroy(s->st);
    delete s;
}

static int stream_create_sized(ffs_ctx* c, uint32_t max_batch, uint32_t cap, uint32_t max_comp, ffs_stream** out);

extern "C" int ffs_stream_create(ffs_ctx* c, ffs_stream** out) {
    if (!c || !out) return FFS_ERR_INVALID;
    return stream_create_sized(c, c->max_batch, c->cap, c->max_comp, out);
}

static int stream_create_sized(ffs_ctx* c, uint32_t max_batch, uint32_t cap, uint32_t max_comp, ffs_stream** out) {
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    ffs_stream* s = new (std::nothrow) ffs_stream();
    if (!s) return FFS_ERR_NOMEM;
    s->ctx = c;
    s->max_batch = max_batch;
    s->cap = cap;
    s->max_comp = max_comp;
    const Layout& L = c->L;
    const size_t B = s->max_batch;
#define STREAM_TRY(expr)                                                        \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            c->err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
            ffs_stream_destroy(s);                                              \
            return e_ == hipErrorOutOfMemory ? FFS_ERR_NOMEM : FFS_ERR_DEVICE;  \
        }                                                                       \
    } while (0)
    // The sparse stages (compaction, union-find, reductions) are latency-bound and keep only a few
    // CUs busy; the streaming candidate kernel is HBM-bound and does not need all 256.  With
    // FFS_CCL_CUS = n > 0 the two groups of kernels run on disjoint CU sets (HIP CU masks), so the
    // connected-components stage of one batch can overlap the threshold stage of the next batch
    // submitted on another ffs_stream.  Measured on MI355X (bench.py, 2 streams): 0 -> 34.4k fps,
    // 32 -> 34.9k, 64 -> 33.8k, 16 -> 21.8k: no gain worth the constraint, so it is off by default.
    {
        const int ncu = c->knobs.ccl_cus;
        hipDeviceProp_t prop;
        STREAM_TRY(hipGetDeviceProperties(&prop, c->device));
        const int total = prop.multiProcessorCount;
        if (ncu > 0 && ncu < total && total <= 1024) {
            const int words = (total + 31) / 32;
            std::vector<uint32_t> m_ccl(words, 0), m_thr(words, 0);
            const int step = total / ncu;  // spread the CCL CUs evenly over the XCDs
            for (int i = 0; i < total; ++i) {
                const bool ccl = (i % step) == 0 && (i / step) < ncu;
                (ccl ? m_ccl : m_thr)[i / 32] |= 1u << (i % 32);
            }
            STREAM_TRY(hipExtStreamCreateWithCUMask(&s->st, (uint32_t)words, m_thr.data()));
            STREAM_TRY(hipExtStreamCreateWithCUMask(&s->st2, (uint32_t)words, m_ccl.data()));
        } else if (c->knobs.sched >= 1) {
            std::lock_guard<std::mutex> lock(c->stream_mu);
            int lo = 0, hi = 0;
            STREAM_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (least, greatest)
            if (c->knobs.sched >= 2) {
                if (!c->dense_st) STREAM_TRY(hipStreamCreateWithPriority(&c->dense_st, hipStreamNonBlocking, c->knobs.sched >= 3 ? (lo + hi) / 2 : lo));
                s->st = c->dense_st;
                if (c->knobs.sched >= 4 && (c->n_streams_made & 1)) {   // (experiment) two dense streams, alternating
                    if (!c->dense_st_b) STREAM_TRY(hipStreamCreateWithPriority(&c->dense_st_b, hipStreamNonBlocking, (lo + hi) / 2));
                    s->st = c->dense_st_b;
                }
                s->st_shared = true;
            } else {
                STREAM_TRY(hipStreamCreateWithPriority(&s->st, hipStreamNonBlocking, lo));
            }
            if (c->knobs.sched >= 2) {
                if (!c->up_st) STREAM_TRY(hipStreamCreateWithFlags(&c->up_st, hipStreamNonBlocking));
                s->st_up = c->up_st;
            }
            if (c->knobs.sched >= 3) {
                const int j = c->n_streams_made++ & 1;
                if (!c->sparse_st[j]) STREAM_TRY(hipStreamCreateWithPriority(&c->sparse_st[j], hipStreamNonBlocking, hi));
                s->st2 = c->sparse_st[j];
                s->st2_shared = true;
            } else {
                STREAM_TRY(hipStreamCreateWithPriority(&s->st2, hipStreamNonBlocking, hi));
            }
        } else {
            STREAM_TRY(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
            s->st2 = s->st;
        }
    }
    if (!s->st_up) s->st_up = s->st;
    {   // more than 64 KB of dynamic LDS has to be asked for, per device
        std::lock_guard<std::mutex> lock(c->stream_mu);
        if (!c->chain_ok) {
            const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_chain<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainDynBytes);
            const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_chain<uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainDynBytes);
            c->chain_ok = e1 == hipSuccess && e2 == hipSuccess;   // (else: the four grid-wide kernels)
            (void)hipGetLastError();
        }
    }
    for (auto& e : s->ev) STREAM_TRY(hipEventCreate(&e));
    STREAM_TRY(hipEventCreate(&s->ev_cs));
    STREAM_TRY(dmalloc(&s->d_img, B * L.frame_stride));
    STREAM_TRY(dmalloc(&s->d_bits, B * L.plane_frame_stride));
    STREAM_TRY(dmalloc(&s->d_sbytes, B * L.bytes_frame_stride));
    STREAM_TRY(dmalloc(&s->d_tile_counts, tile_counts_bytes(s)));
    STREAM_TRY(dmalloc(&s->d_occ, B * (size_t)occ_frame_words(L) * 4));
    STREAM_TRY(dmalloc(&s->d_bright, (size_t)kBrightCap * sizeof(uint2)));

    // per-frame counters in the layout of h_counts, so that one copy brings them all back:
    // [B] strong pixels | [B] components | [B][8] summary | [1] overflow / error flag
    STREAM_TRY(dmalloc(&s->d_num_strong, (B * 10 + 1) * 4));
    s->d_n_comp = s->d_num_strong + B;
    s->d_summary = s->d_num_strong + 2 * B;
    s->d_overflow = s->d_num_strong + 10 * B;
    STREAM_TRY(dmalloc(&s->d_row_off, B * (size_t)(L.H + 1) * 4));
    STREAM_TRY(dmalloc(&s->d_list_k, B * (size_t)s->cap * 4));
    STREAM_TRY(dmalloc(&s->d_list_i, B * (size_t)s->cap * 4));
    STREAM_TRY(dmalloc(&s->d_parent, B * (size_t)s->cap * 4));
    STREAM_TRY(dmalloc(&s->d_comp_id, B * (size_t)s->cap * 4));
    STREAM_TRY(dmalloc(&s->d_part_roots, B * (size_t)kLabelParts * 4));
    STREAM_TRY(dmalloc(&s->d_acc, B * (size_t)s->max_comp * sizeof(CompAcc)));
    STREAM_TRY(dmalloc(&s->d_acc2, B * (size_t)s->cap * sizeof(CompAcc2)));
    STREAM_TRY(dmalloc(&s->d_chunk_roots, B * (size_t)(s->cap / kRootChunk + 1) * 4));
    STREAM_TRY(dmalloc(&s->d_recs, B * (size_t)s->max_comp * sizeof(ReflOut)));
    // raw frames, or bitshuffle-LZ4 chunks (which can exceed the raw size by < 1 % when incompressible)
    s->h_img_bytes = B * ((size_t)L.W * L.H * c->pixel_bytes + (size_t)L.W * L.H * c->pixel_bytes / 128 + 4096);
    STREAM_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_img), s->h_img_bytes, hipHostMallocDefault));
    STREAM_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_counts), (B * 11 + 1) * 4, hipHostMallocDefault));  // (+ [B] per-frame flags, k_frame_chain)
    std::memset(s->h_counts, 0, (B * 11 + 1) * 4);
    STREAM_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_recs), B * (size_t)s->max_comp * sizeof(ReflOut),
                             hipHostMallocDefault));
    // k_finalize can write the (few MB of) records straight into this pinned, device-visible buffer:
    // no copy kernel after it.  FFS_DIRECT_RECS=0 keeps the device buffer + copy (A/B).
    s->direct_recs = c->knobs.direct_recs != 0;
    if (s->direct_recs
        && hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_recs_dev), s->h_recs, 0) != hipSuccess) {
        (void)hipGetLastError();
        s->direct_recs = false;
    }
    if (s->direct_recs && hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_counts_dev), s->h_counts, 0) != hipSuccess) {
        (void)hipGetLastError();
        s->h_counts_dev = nullptr;
    }
    STREAM_TRY(hipMemsetAsync(s->d_overflow, 0, 4, s->st));
    STREAM_TRY(hipMemsetAsync(s->d_occ, 0, B * (size_t)occ_frame_words(L) * 4, s->st));
    // bits beyond the image width (x >= W up to the row pitch) are never written by the threshold kernels and must read 0
    STREAM_TRY(hipMemsetAsync(s->d_bits, 0, B * L.plane_frame_stride, s->st));
    STREAM_TRY(hipStreamSynchronize(s->st));
#undef STREAM_TRY
    *out = s;
    return FFS_OK;
}

extern "C" int ffs_stream_host_buffer(ffs_stream* s, void** ptr, size_t* bytes) {
    if (!s) return FFS_ERR_INVALID;
    if (ptr) *ptr = s->h_img;
    if (bytes) *bytes = s->h_img_bytes;
    return FFS_OK;
}

static ThresholdArgs make_threshold_args(ffs_stream* s, const void* img, size_t pitch, size_t fstride,
                                         uint32_t n_frames) {
    const ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    const ffs_params& p = s->batch_params;
    ThresholdArgs a{};
    a.image = img;
    a.frame_stride = fstride;
    a.pitch = (uint32_t)pitch;
    a.maskbits = c->d_maskbits;
    a.bits = s->d_bits;
    a.strong_bytes = s->d_sbytes;
    a.tile_counts = s->d_tile_counts;
    a.W = L.W;
    a.H = L.H;
    a.pitch_px = L.pitch_px;
    a.mpitch = L.mpitch;
    a.bpitch = L.bpitch;
    a.plane_frame_stride = L.plane_frame_stride;
    a.bytes_frame_stride = L.bytes_frame_stride;
    a.n_strips = c->n_strips;
    a.n_tiles = c->n_tiles;
    // Bands: enough waves to fill the 256 CUs several times over, bands no shorter than 24 rows -- and a
    // whole number of bands per XCD.  The candidate kernels deal the bands round-robin to the 8 XCDs
    // (band = xcd + 8 k, so that neighbouring strips share an L2); with 29 bands three XCDs had a band
    // less to do than the others and the launch waited for the busy five: 511 us per 32 Eiger frames, against 430-440 us with 48 or 56.
    const long long target_waves = c->knobs.target_waves;
    {
        const long long per_band = std::max<long long>(1, (long long)c->n_strips * n_frames);
        // (bands of at least 72 rows keep the 6-row warm-up of every band below 8 %)
        long long nb = std::max<long long>(1, std::min<long long>(target_waves / per_band, L.H / 72));
        if (nb >= 8) nb = nb / 8 * 8;
        a.band_rows = (int)std::min<long long>(512, (L.H + nb - 1) / nb);
        a.n_bands = (L.H + a.band_rows - 1) / a.band_rows;
    }
    a.kS = (float)(p.nsig_s * p.nsig_s * (1.0 - 1.0 / 65536.0));
    a.kB = (float)(p.nsig_b * (1.0 - 1.0 / 1048576.0));
    a.min_count = p.min_count;
    a.nsig_b = p.nsig_b;
    a.nsig_s = p.nsig_s;
    a.nsig_b2 = p.nsig_b * p.nsig_b;
    a.nsig_s2 = p.nsig_s * p.nsig_s;
    a.threshold = p.threshold;
    a.max_valid = p.max_valid;
    {   // FFS_K1_VARIANT (A/B testing): 0 = signal-test-only candidate kernel + exact kernel, 1 = signal +
        // dispersion screen in the candidate kernel + exact kernel; default 2 (16-bit pixels) = the whole
        // threshold in one streaming kernel (kernels_stream.hpp)
        a.variant = c->knobs.k1_variant;
        if (s->force_variant >= 0) a.variant = std::min(a.variant, s->force_variant);
    }
    a.overflow = s->d_overflow;
    a.bright_n = s->d_tile_counts + tile_counts_bytes(s) / 4 - 1;
    a.bright_list = s->d_bright;
    a.bright_cap = std::min<uint32_t>(kBrightCap, (uint32_t)c->knobs.bright_cap);
    a.dbg = c->knobs.k1_debug;
    // The byte mask (the reference kernel's result_strong, 1 byte per pixel) is an OUTPUT only when it was asked for
    // (want_strong_mask: --writeout, parity tests): the hot path's own strong mask is the bit plane, and the 0.58 GB of
    // zeros per 32 Eiger frames cost the streaming kernel 15 % (FFS_DENSE_MASK=1 forces them, for A/B and the roofline leg)
    a.occ = s->d_occ;
    a.occ_frame_words = occ_frame_words(L);
    a.occ_spr = L.mpitch / 16;
    a.dense_mask = (p.want_strong_mask || c->knobs.dense_mask || (a.dbg & 16)) && !(a.dbg & 8) ? 1 : 0;
    a.ginfo = c->d_ginfo;
    a.mmap = c->d_mmap;
    a.gpitch = (uint32_t)L.pitch_px * (uint32_t)c->pixel_bytes / 4;
    a.gpf = c->pixel_bytes == 2 ? (L.W + 7) / 8 : (L.W + 3) / 4;
    a.n_frames = (int)n_frames;
    {   // frames side by side in one super row, as many as keep every buffer of the group below 2 GiB
        const uint64_t per_frame = std::max<uint64_t>(fstride, L.bytes_frame_stride);
        a.group_frames = (int)std::max<uint64_t>(1, std::min<uint64_t>(n_frames, ((1ull << 31) - 1) / per_frame));
        a.group_frames = std::min(a.group_frames, c->knobs.k1_group);
        const int n_groups = ((int)n_frames + a.group_frames - 1) / a.group_frames;
        const long long lanes = (long long)a.group_frames * (a.gpf + 1);
        const long long lines = (long long)a.group_frames * (L.bpitch / 128);  // byte-mask lines to zero per row
        const int lines_per_wave = c->pixel_bytes == 2 ? 4 : 2;  // a wave zero-fills 512 / 256 bytes of the byte mask per row
        a.s_strips = (int)std::max<long long>((lanes + kSOwned - 1) / kSOwned, (lines + lines_per_wave - 1) / lines_per_wave);
        const long long tw = c->knobs.target_waves;
        const long long per_band = std::max<long long>(1, (long long)a.s_strips * n_groups);
        long long nb = std::max<long long>(1, std::min<long long>(tw / per_band, L.H / 72));
        if (nb >= 8) nb = nb / 8 * 8;
        a.s_band_rows = (int)std::min<long long>(1024, (L.H + nb - 1) / nb);
        a.s_bands = (L.H + a.s_band_rows - 1) / a.s_band_rows;
    }
    a.dplane = s->d_dplane;
    a.eplane = s->d_eplane;
    a.ext_flavour = p.extended_flavour;
    {
        a.ext_variant = c->knobs.ext_variant;
    }
    a.ext_strips = (L.pitch_px + kExtOwnedPx - 1) / kExtOwnedPx;
    {   // one pixel per lane: bands of 64..256 rows keep the 6-row warm-up below 10 %
        const long long ext_target = 8192;
        long long er = ((long long)L.H * a.ext_strips * n_frames + 4 * ext_target - 1) / (4 * ext_target);
        er = std::max<long long>(64, std::min<long long>(er, 256));
        a.ext_band_rows = (int)er;
        a.ext_bands = (L.H + a.ext_band_rows - 1) / a.ext_band_rows;
    }
    return a;
}

// Extended dispersion: first pass -> erosion -> final threshold (kernels_extended.hpp).  Leaves the
// strong plane in a.bits, the byte mask and the per-tile counts exactly as launch_exact does.
// First pass.  16-bit pixels: the streaming candidate kernel in its extended mode + the exact
// dispersion test on its candidates (ext_variant 1, default); k_ext_first is the plain one-pixel-
// per-lane kernel that computes the same plane directly (32-bit pixels, and FFS_EXT_VARIANT=0).
static bool ext_fast_first(const ffs_stream* s, const ThresholdArgs& a) {
    return s->ctx->pixel_bytes == 2 && a.ext_variant >= 1;
}
// FFS_EXT_VARIANT (16-bit pixels): 2 (default) = k_stream_u16<.., true>, the streaming kernel deciding the first pass
// exactly in its drain; 1 = round 1's candidate kernel in extended mode + exact kernel; 0 = k_ext_first.
static bool ext_stream_first(const ffs_stream* s, const ThresholdArgs& a) {
    return s->ctx->pixel_bytes == 2 && a.ext_variant >= 2;
}
static void launch_ext_first(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames) {
    if (ext_stream_first(s, a)) {
        ThresholdArgs b = a;
        b.n_strips = a.s_strips;
        b.band_rows = a.s_band_rows;
        b.n_bands = a.s_bands;
        // the kernel writes the non-zero bytes of the first-pass plane; the bright-list count sits behind the tile counts
        (void)hipMemsetAsync(a.dplane, 0, (size_t)n_frames * a.plane_frame_stride, s->st);
        (void)hipMemsetAsync(a.tile_counts, 0, tile_counts_bytes(s), s->st);
        const int bands8s = (b.n_bands + 7) / 8 * 8;
        const unsigned n_groups = (n_frames + (unsigned)a.group_frames - 1) / (unsigned)a.group_frames;
        hipLaunchKernelGGL((k_stream_u16<2, true>), dim3((unsigned)(b.n_strips * bands8s), n_groups), dim3(64), 0, s->st, b);
        hipLaunchKernelGGL((k_bright_fix<uint16_t, true>), dim3(32), dim3(256), 0, s->st, b);
        return;
    }
    if (ext_fast_first(s, a)) {
        // a.bits collects the positives the streaming kernel could not settle: zero it, the drain fills it
        (void)hipMemsetAsync(a.bits, 0, (size_t)n_frames * a.plane_frame_stride, s->st);
        const int bands8 = (a.n_bands + 7) / 8 * 8;
        hipLaunchKernelGGL((k_candidates_u16<true, true>), dim3((unsigned)(a.n_strips * bands8), n_frames), dim3(64), 0,
                           s->st, a);
        return;
    }
    dim3 g1((unsigned)(a.ext_strips * a.ext_bands), n_frames);
    if (s->ctx->pixel_bytes == 2) hipLaunchKernelGGL(k_ext_first<uint16_t>, g1, dim3(64), 0, s->st, a);
    else hipLaunchKernelGGL(k_ext_first<uint32_t>, g1, dim3(64), 0, s->st, a);
}
static void launch_ext_rest(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames) {
    const bool u16 = s->ctx->pixel_bytes == 2;
    if (ext_stream_first(s, a)) {
        // (the streaming kernel zero-filled the byte mask and left the exact first-pass plane)
    } else if (ext_fast_first(s, a))  // (the candidate kernel also zero-filled the byte mask)
        hipLaunchKernelGGL(k_exact_disp<uint16_t>, dim3((unsigned)a.n_tiles, n_frames), dim3(256), 0, s->st, a);
    else
        (void)hipMemsetAsync(a.strong_bytes, 0, (size_t)n_frames * a.bytes_frame_stride, s->st);
    const unsigned erode_lanes = (a.mpitch / 4) * (unsigned)((a.H + kErodeRows - 1) / kErodeRows);
    hipLaunchKernelGGL(k_ext_erode, dim3((erode_lanes + 255) / 256, n_frames), dim3(256), 0, s->st, a);
    dim3 g3((unsigned)a.n_tiles, n_frames);
    if (u16) hipLaunchKernelGGL(k_ext_final<uint16_t>, g3, dim3(256), 0, s->st, a);
    else hipLaunchKernelGGL(k_ext_final<uint32_t>, g3, dim3(256), 0, s->st, a);
}
static void launch_extended(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames) {
    launch_ext_first(s, a, n_frames);
    launch_ext_rest(s, a, n_frames);
}

static int ensure_extended_buffers(ffs_stream* s) {
    if (s->d_dplane) return FFS_OK;
    ffs_ctx* c = s->ctx;
    const size_t bytes = (size_t)s->max_batch * c->L.plane_frame_stride;
    if (dmalloc(&s->d_dplane, bytes) != hipSuccess || dmalloc(&s->d_eplane, bytes) != hipSuccess) {
        (void)hipGetLastError();
        c->err = "hipMalloc(extended dispersion planes) failed";
        return FFS_ERR_NOMEM;
    }
    return FFS_OK;
}

// fix_st: the stream of the bright-window fix-up that follows a streaming kernel (nullptr: the dense stream itself)
static void launch_candidates(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames, hipStream_t fix_st = nullptr, hipEvent_t fix_after = nullptr,
                              bool skip_fix = false) {
    if (a.variant >= 2) {
        // the whole threshold in one kernel: final strong plane + per-tile counts (atomics into zeroed counters)
        ThresholdArgs b = a;
        b.n_strips = a.s_strips;
        b.band_rows = a.s_band_rows;
        b.n_bands = a.s_bands;
        // (per-tile counts and the bright-list count are zero here: k_union of the previous batch cleared them, or enqueue_batch did)
        const int bands8s = (b.n_bands + 7) / 8 * 8;
        const unsigned n_groups = (n_frames + (unsigned)a.group_frames - 1) / (unsigned)a.group_frames;
        const int ahead = s->ctx->knobs.k1_ahead;
        if (s->ctx->pixel_bytes == 4) {
            if (fix_st && s->ctx->knobs.ext_launch) {
                hipExtLaunchKernelGGL(k_stream_u32<2>, dim3((unsigned)(b.n_strips * bands8s), n_groups), dim3(64), 0, s->st,
                                      s->ev1_pending ? s->ev[1] : nullptr, fix_after, 0, b);
                s->ev1_pending = false;
                (void)hipStreamWaitEvent(fix_st, fix_after, 0);
                if (!skip_fix) hipLaunchKernelGGL(k_bright_fix<uint32_t>, dim3(32), dim3(256), 0, fix_st, b);
                return;
            }
            hipLaunchKernelGGL(k_stream_u32<2>, dim3((unsigned)(b.n_strips * bands8s), n_groups), dim3(64), 0, s->st, b);
            if (fix_st) {
                (void)hipEventRecord(fix_after, s->st);
                (void)hipStreamWaitEvent(fix_st, fix_after, 0);
            }
            hipLaunchKernelGGL(k_bright_fix<uint32_t>, dim3(32), dim3(256), 0, fix_st ? fix_st : s->st, b);
            return;
        }
        if (ahead >= 3)
            hipLaunchKernelGGL(k_stream_u16<3>, dim3((unsigned)(b.n_strips * bands8s), n_groups), dim3(64), 0, s->st, b);
        else if (fix_st && s->ctx->knobs.ext_launch) {
            // start and stop events ride on the dispatch itself (its completion signal): no marker packets around it
            hipExtLaunchKernelGGL(k_stream_u16<2>, dim3((unsigned)(b.n_strips * bands8s), n_groups), dim3(64), 0, s->st,
                                  s->ev1_pending ? s->ev[1] : nullptr, fix_after, 0, b);
            s->ev1_pending = false;
            (void)hipStreamWaitEvent(fix_st, fix_after, 0);
            if (!skip_fix) hipLaunchKernelGGL(k_bright_fix<uint16_t>, dim3(32), dim3(256), 0, fix_st, b);
            return;
        } else
            hipLaunchKernelGGL(k_stream_u16<2>, dim3((unsigned)(b.n_strips * bands8s), n_groups), dim3(64), 0, s->st, b);
        if (fix_st) {
            (void)hipEventRecord(fix_after, s->st);
            (void)hipStreamWaitEvent(fix_st, fix_after, 0);
        }
        hipLaunchKernelGGL(k_bright_fix<uint16_t>, dim3(32), dim3(256), 0, fix_st ? fix_st : s->st, b);
        return;
    }
    const int bands8 = (a.n_bands + 7) / 8 * 8;  // XCD-aware mapping wants a multiple of 8 bands
    dim3 grid((unsigned)(a.n_strips * bands8), n_frames), block(64);
    if (s->ctx->pixel_bytes == 2) {
        if (a.variant == 0)
            hipLaunchKernelGGL(k_candidates_u16<false>, grid, block, 0, s->st, a);
        else
            hipLaunchKernelGGL(k_candidates_u16<true>, grid, block, 0, s->st, a);
    }
    else if (a.variant >= 1) {
        // the queue variant ORs its few candidate nibbles into a zeroed plane (a lane pair shares a byte)
        (void)hipMemsetAsync(a.bits, 0, (size_t)n_frames * a.plane_frame_stride, s->st);
        hipLaunchKernelGGL(k_candidates_u32_q, grid, block, 0, s->st, a);
    } else
        hipLaunchKernelGGL(k_candidates_u32, grid, block, 0, s->st, a);
}

static void launch_exact(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames) {
    if (a.variant >= 2) return;  // k_stream_u16 / k_stream_u32 left the final plane and the counts
    dim3 grid((unsigned)a.n_tiles, n_frames), block(256);
    if (s->ctx->pixel_bytes == 2 && a.variant == 1)  // few candidates per tile: one wave per tile
        hipLaunchKernelGGL(k_exact_w64<uint16_t>, grid, dim3(64), 0, s->st, a);
    else if (s->ctx->pixel_bytes == 2)
        hipLaunchKernelGGL(k_exact<uint16_t>, grid, block, 0, s->st, a);
    else if (a.variant >= 1)
        hipLaunchKernelGGL(k_exact_w64<uint32_t>, grid, dim3(64), 0, s->st, a);
    else
        hipLaunchKernelGGL(k_exact<uint32_t>, grid, block, 0, s->st, a);
}

static int check_layout(ffs_stream* s, size_t pitch, size_t fstride, uint32_t n_frames) {
    ffs_ctx* c = s->ctx;
    if (n_frames == 0 || n_frames > s->max_batch) {
        c->err = "n_frames must be in 1..max_batch";
        return FFS_ERR_INVALID;
    }
    if (pitch % 16 || pitch < (size_t)c->L.pitch_px * c->pixel_bytes || pitch >= (1ull << 32)
        || fstride < pitch * c->L.H || (pitch * c->L.H) >= (1ull << 32)) {
        c->err = "device layout: pitch must be a multiple of 16 bytes and >= round_up(width,128)*pixel_bytes; "
                 "frame_stride >= pitch*height";
        return FFS_ERR_INVALID;
    }
    return FFS_OK;
}

// (experiment) occupies slots for a given time without touching memory
__global__ void k_dummy_spin(uint32_t ticks, uint32_t* sink) {
    extern __shared__ uint32_t s_dummy[];
    const uint64_t t0 = wall_clock64();
    uint32_t it = 0;
    while (wall_clock64() - t0 < ticks && it < (1u << 20)) { __builtin_amdgcn_s_sleep(20); ++it; }
    if (it == 0xFFFFFFFFu) { s_dummy[threadIdx.x] = it; *sink = s_dummy[0]; }
}

static int enqueue_batch(ffs_stream* s, const void* d_img, size_t pitch, size_t fstride, uint32_t n,
                         const ffs_params* snapshot = nullptr) {
    ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    s->batch_params = snapshot ? *snapshot : c->params;
    const ffs_params& p = s->batch_params;
    s->cur_img = d_img;
    s->cur_pitch = pitch;
    s->cur_fstride = fstride;

    (void)hipGetLastError();  // drop any stale error state: the check below is for OUR launches
    if (s->st_up != s->st && !s->dev_input) HIP_TRY(c, hipStreamWaitEvent(s->st, s->ev[1], 0));   // the frames are in place (upload / decode stream)
    if (p.algorithm == FFS_ALGO_DISPERSION_EXTENDED) {
        const int rc = ensure_extended_buffers(s);
        if (rc != FFS_OK) return rc;
    }
    const ThresholdArgs ta = make_threshold_args(s, d_img, pitch, fstride, n);
    // (Measured and dropped: chaining the dense kernels of different streams with events so that one
    // stream's sparse stage runs under the other's dense kernel -- a small kernel queued behind a
    // 9000-workgroup dispatch of another queue gets no CUs until that dispatch drains; 35.3 k vs
    // 37.2 k frames/s.  CU masks for the two stages: no gain either.)
    const bool one_kernel = ta.variant >= 2 && p.algorithm != FFS_ALGO_DISPERSION_EXTENDED;
    if (one_kernel && s->bits_dirty)  // (another algorithm / variant or a failed batch left bits behind)
        HIP_TRY(c, hipMemsetAsync(s->d_bits, 0, (size_t)s->max_batch * L.plane_frame_stride, s->st));
    if (one_kernel && s->counts_dirty)
        HIP_TRY(c, hipMemsetAsync(s->d_tile_counts, 0, tile_counts_bytes(s), s->st));
    s->counts_dirty = true;
    s->bits_dirty = true;  // until every launch of this batch is enqueued (a failure in between leaves bits behind)
    const bool will_ext_launch = one_kernel && s->st2 != s->st && c->knobs.fix_aside && c->knobs.ext_launch && c->knobs.k1_ahead < 3;
    // FFS_CCL = 2 (default): the whole sparse stage in one launch, one workgroup per frame (kernels_chain.hpp)
    const bool will_chain = c->knobs.ccl_variant >= 2 && L.H <= 65535 && c->chain_ok && s->direct_recs && s->h_counts_dev
                            && c->n_tiles <= kChainMaxTiles && L.H <= kChainMaxRows && !c->knobs.chain_skip;
    // ... which then also does the bright-window fix-up, and whose workgroups (a whole CU each) should get their CUs
    // BEFORE the next batch's streaming kernel floods the dispatcher: that kernel waits for this launch to have STARTED.
    // (Without it a batch's sparse launch sits out the whole next streaming kernel: 0.35 ms more latency per batch.)
    // Only while few batches are in flight: with a deep pipeline the latency is hidden anyway, the wait costs the dense
    // stream ~15 us per batch and the fix-up inside the one-workgroup-per-frame launch ~25 us of its CUs (4 batches in
    // flight: 0.369-0.377 against 0.353 ms per step; 2 in flight: 0.385 against 0.523).
    const int depth = c->inflight.load() + (s->busy ? 0 : 1);
    const bool chain_first = one_kernel && will_chain && will_ext_launch && c->knobs.chain_first > 0 && depth <= c->knobs.chain_first;
    const bool fold_fix = chain_first;
    if (chain_first) {
        hipEvent_t prev = c->last_chain_start.load();
        if (prev) HIP_TRY(c, hipStreamWaitEvent(s->st, prev, 0));
    }
    if (s->ev1_pending && !will_ext_launch) {
        HIP_TRY(c, hipEventRecord(s->ev[1], s->st));
        s->ev1_pending = false;
    }
    if (p.algorithm == FFS_ALGO_DISPERSION_EXTENDED) {
        launch_extended(s, ta, n);
    } else {
        // with a sparse stream of its own, the bright-window fix-up goes there: the dense stream holds streaming
        // kernels only, back to back
        const bool fix_aside = one_kernel && s->st2 != s->st && c->knobs.fix_aside;
        launch_candidates(s, ta, n, fix_aside ? s->st2 : nullptr, s->ev[2], fold_fix);
        launch_exact(s, ta, n);
        if (fix_aside) {   // (ev[2] was recorded behind the streaming kernel, and st2 waits for it already)
            HIP_TRY(c, hipGetLastError());
            goto dense_done;
        }
    }
    HIP_TRY(c, hipEventRecord(s->ev[2], s->st));
    if (s->st2 != s->st) HIP_TRY(c, hipStreamWaitEvent(s->st2, s->ev[2], 0));
dense_done:

    CclArgs ca{};
    ca.image = d_img;
    ca.frame_stride = fstride;
    ca.pitch = (uint32_t)pitch;
    ca.bits = s->d_bits;
    ca.clear_bits = (ta.variant >= 2 && p.algorithm != FFS_ALGO_DISPERSION_EXTENDED) ? 1 : 0;
    s->bits_cleared = ca.clear_bits != 0;
    ca.tile_counts = s->d_tile_counts;
    ca.num_strong = s->d_num_strong;
    ca.row_off = s->d_row_off;
    ca.list_k = s->d_list_k;
    ca.list_i = s->d_list_i;
    ca.parent = s->d_parent;
    ca.comp_id = s->d_comp_id;
    ca.n_comp = s->d_n_comp;
    ca.overflow = s->d_overflow;
    ca.W = L.W;
    ca.H = L.H;
    ca.pitch_px = L.pitch_px;
    ca.mpitch = L.mpitch;
    ca.plane_frame_stride = L.plane_frame_stride;
    ca.n_tiles = c->n_tiles;
    ca.cap = s->cap;
    ca.max_comp = s->max_comp;
    ca.pixel_bytes = c->pixel_bytes;
    ca.strong_bytes = s->d_sbytes;
    ca.bpitch = L.bpitch;
    ca.bytes_frame_stride = L.bytes_frame_stride;
    // FFS_CCL (A/B): 1 (default) = accumulators at the root, 4 launches (emit, union, reduce, finalize) and 40-byte
    // records on the wire; 0 = numbered components (count, label, reduce, finalize), 56-byte records
    const int ccl_variant = c->knobs.ccl_variant;
    const bool root_mode = ccl_variant >= 1 && L.H <= 65535;
    ca.acc2 = root_mode ? s->d_acc2 : nullptr;
    ca.n_comp = s->d_n_comp;
    ca.summary = s->d_summary;
    s->wire2 = root_mode;
    // (Measured and dropped: one workgroup per frame with the union-find forest, the entries' columns and
    // the row offsets in LDS instead of k_union + k_label -- correct, but 64 us against 50 + 24 us: a
    // frame's ~18 k entries are compute-bound on a single CU.)
    // (Measured and dropped: writing the list from the exact stage itself, each tile getting its list
    // offset by a decoupled look-back over the tiles before it -- 263 us against 87 + 4 + 63 us for
    // the then three kernels: tiles that wait for a predecessor's count hold their CU slots.)
    // FFS_EMIT (A/B): 1 (default) = one wave per tile, runs linked in the same pass; 0 = one workgroup per tile + k_link_runs
    const int emit_variant = c->knobs.emit_variant;
    const int skip = c->knobs.chain_skip;
    // (the kernels that zero-fill the mask: streaming kernels only when asked; the older threshold kernels and the
    // extended algorithm's last pass always do)
    ca.dense_bytes = ((one_kernel || (p.algorithm == FFS_ALGO_DISPERSION_EXTENDED && ext_stream_first(s, ta))) ? ta.dense_mask : 1) ? 1 : 0;
    s->dense_valid = ca.dense_bytes != 0;
    ca.occ = s->d_occ;
    ca.occ_frame_words = occ_frame_words(L);
    ca.occ_spr = L.mpitch / 16;
    ca.use_occ = (one_kernel && c->knobs.use_occ) ? 1 : 0;   // (only the streaming kernels and their fix-up keep the bitmap)
    // FFS_CCL = 2 (default): the whole sparse stage in one launch, one workgroup per frame (kernels_chain.hpp)
    s->chain_mode = root_mode && will_chain && !skip;
    if (s->chain_mode) {
        ChainArgs A{};
        A.c = ca;
        SegArgs& sa = A.s;
        sa.list_k = s->d_list_k;
        sa.list_i = s->d_list_i;
        sa.parent = s->d_parent;
        sa.seg_n = s->d_num_strong;
        sa.seg_stride = s->cap;
        sa.n_comp = s->d_n_comp;
        sa.max_comp = s->max_comp;
        sa.overflow = s->d_overflow;
        sa.W = (uint32_t)L.W;
        sa.H = (uint32_t)L.H;
        sa.row_off = s->d_row_off;
        sa.n_slices = 1;
        sa.min_spot_size = p.min_spot_size;
        sa.max_sep = p.max_peak_centroid_separation;
        sa.recs = s->h_recs_dev;
        sa.summary = s->d_summary;
        sa.runs_linked = 2;
        sa.acc2 = s->d_acc2;
        sa.zero_counts = s->d_tile_counts;
        sa.zero_per_seg = (uint32_t)c->n_tiles;
        sa.zero_word = s->d_tile_counts + tile_counts_bytes(s) / 4 - 1;
        A.h_counts = s->h_counts_dev;
        A.max_batch = (uint32_t)s->max_batch;
        A.rec_stride = s->max_comp;
        A.stop_after = c->knobs.chain_stop;
        A.t = ta;
        A.fix_bright = fold_fix ? 1 : 0;
        A.fix_done = s->d_tile_counts + tile_counts_bytes(s) / 4 - 2;
        if (c->pixel_bytes == 2) hipExtLaunchKernelGGL(k_frame_chain<uint16_t>, dim3(n), dim3(kChainThreads), kChainDynBytes, s->st2, s->ev_cs, nullptr, 0, A);
        else hipExtLaunchKernelGGL(k_frame_chain<uint32_t>, dim3(n), dim3(kChainThreads), kChainDynBytes, s->st2, s->ev_cs, nullptr, 0, A);
        if (one_kernel && s->st2 != s->st) c->last_chain_start.store(s->ev_cs);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(s->ev[4], s->st2));
        s->ev3_is_ev4 = true;
        s->spec_recs_copied = (uint64_t)s->max_batch * s->max_comp;
        s->bits_dirty = !one_kernel;
        s->counts_dirty = false;
        mark_busy(s);
        s->n_frames = n;
        return FFS_OK;
    }
    if (skip & 1) {
        const int us = c->knobs.dummy_us;
        if (us > 0)
            hipLaunchKernelGGL(k_dummy_spin, dim3(c->knobs.dummy_wg), dim3(c->knobs.dummy_threads),
                               (size_t)c->knobs.dummy_lds, s->st2, (uint32_t)us * 100u, s->d_tile_counts);
    } else if (emit_variant >= 1 || root_mode) {
        if (c->pixel_bytes == 2)
            hipLaunchKernelGGL(k_emit_list_w<uint16_t>, dim3(c->n_tiles, n), dim3(64), 0, s->st2, ca);
        else
            hipLaunchKernelGGL(k_emit_list_w<uint32_t>, dim3(c->n_tiles, n), dim3(64), 0, s->st2, ca);
    } else if (c->pixel_bytes == 2)
        hipLaunchKernelGGL(k_emit_list<uint16_t>, dim3(c->n_tiles, n), dim3(256), 0, s->st2, ca);
    else
        hipLaunchKernelGGL(k_emit_list<uint32_t>, dim3(c->n_tiles, n), dim3(256), 0, s->st2, ca);

    SegArgs sa{};
    sa.list_k = s->d_list_k;
    sa.list_i = s->d_list_i;
    sa.parent = s->d_parent;
    sa.comp_id = s->d_comp_id;
    sa.seg_n = s->d_num_strong;
    sa.seg_stride = s->cap;
    sa.n_comp = s->d_n_comp;
    sa.acc = s->d_acc;
    sa.max_comp = s->max_comp;
    sa.overflow = s->d_overflow;
    sa.W = (uint32_t)L.W;
    sa.H = (uint32_t)L.H;
    sa.row_off = s->d_row_off;
    sa.slice_begin = nullptr;
    sa.n_slices = 1;
    sa.min_spot_size = p.min_spot_size;
    sa.max_sep = p.max_peak_centroid_separation;
    sa.recs = s->direct_recs ? s->h_recs_dev : s->d_recs;
    sa.summary = s->d_summary;
    const int gx = c->knobs.ccl_grid;
    const dim3 gseg((unsigned)gx, n), b256(256);
    const int link_runs = c->knobs.link_runs;
    sa.runs_linked = (emit_variant >= 1 || root_mode) ? 2 : link_runs;
    sa.acc2 = s->d_acc2;
    sa.chunk_roots = s->d_chunk_roots;
    sa.chunks_max = s->cap / kRootChunk + 1;
    sa.zero_counts = s->d_tile_counts;
    sa.zero_per_seg = (uint32_t)c->n_tiles;
    sa.zero_word = s->d_tile_counts + tile_counts_bytes(s) / 4 - 1;
    if (emit_variant < 1 && !root_mode && link_runs) hipLaunchKernelGGL(k_link_runs, gseg, b256, 0, s->st2, sa);
    if (!(skip & 3)) hipLaunchKernelGGL(k_union<false>, gseg, b256, 0, s->st2, sa);
    if (skip) {
        if (!(skip & 7)) hipLaunchKernelGGL(k_reduce_roots, gseg, b256, 0, s->st2, sa);
    } else if (root_mode) {
        hipLaunchKernelGGL(k_reduce_roots, gseg, b256, 0, s->st2, sa);
        hipLaunchKernelGGL(k_finalize_roots, gseg, b256, 0, s->st2, sa);
    } else {
        sa.part_roots = s->d_part_roots;
        hipLaunchKernelGGL(k_count_roots, dim3(kLabelParts, n), b256, 0, s->st2, sa);
        hipLaunchKernelGGL(k_label_parts, dim3(kLabelParts, n), b256, 0, s->st2, sa);
        hipLaunchKernelGGL(k_reduce<false>, gseg, b256, 0, s->st2, sa);
        hipLaunchKernelGGL(k_finalize<false>, dim3(8, n), b256, 0, s->st2, sa);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(s->ev[3], s->st2));
    s->ev3_is_ev4 = false;

    // small counts first; ffs_wait() sizes the record copy from them
    const size_t B = s->max_batch;
    HIP_TRY(c, hipMemcpyAsync(s->h_counts, s->d_num_strong, (B * 10 + 1) * 4, hipMemcpyDeviceToHost, s->st2));
    if (s->direct_recs) {
        s->spec_recs_copied = (uint64_t)B * s->max_comp;  // everything is on the host already
    } else {
        s->spec_recs_copied = std::min<uint64_t>((uint64_t)s->spec_recs_per_frame * n, (uint64_t)B * s->max_comp);
        HIP_TRY(c, hipMemcpyAsync(s->h_recs, s->d_recs, s->spec_recs_copied * (s->wire2 ? sizeof(WireRec2) : sizeof(ReflOut)),
                                  hipMemcpyDeviceToHost, s->st2));
    }
    HIP_TRY(c, hipEventRecord(s->ev[4], s->st2));
    s->bits_dirty = !one_kernel;  // the compaction of a one-kernel batch leaves the plane all zero again
    s->counts_dirty = (skip & 3) != 0;  // k_union cleared the counts of the frames of this batch (all the streaming kernel touched)
    mark_busy(s);
    s->n_frames = n;
    return FFS_OK;
}

extern "C" int ffs_submit_device(ffs_stream* s, const void* device_pixels, size_t pitch, size_t fstride,
                                 uint32_t n_frames, int64_t first_frame_id) {
    if (!s || !device_pixels) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream already has a batch in flight: call ffs_wait() first";
        return FFS_ERR_INVALID;
    }
    int rc = check_layout(s, pitch, fstride, n_frames);
    if (rc != FFS_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    // (no marker here: every packet in the dense stream is ~5 us between two streaming kernels; enqueue_batch attaches
    // the start event to its first kernel where it can, or records it)
    s->dev_input = true;
    s->ev1_pending = true;
    s->first_id = first_frame_id;
    return enqueue_batch(s, device_pixels, pitch, fstride, n_frames);
}

extern "C" int ffs_submit(ffs_stream* s, const void* host_pixels, uint32_t n_frames, int64_t first_frame_id) {
    if (!s || !host_pixels) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream already has a batch in flight: call ffs_wait() first";
        return FFS_ERR_INVALID;
    }
    if (n_frames == 0 || n_frames > s->max_batch) {
        c->err = "n_frames must be in 1..max_batch";
        return FFS_ERR_INVALID;
    }
    const Layout& L = c->L;
    HIP_TRY(c, hipSetDevice(c->device));
    s->dev_input = false;
    HIP_TRY(c, hipEventRecord(s->ev[0], s->st_up));
    // one 2D copy: the default device layout keeps frames contiguous (frame_stride = H * pitch)
    const size_t row = (size_t)L.W * c->pixel_bytes;
    HIP_TRY(c, hipMemcpy2DAsync(s->d_img, L.pitch, host_pixels, row, row, (size_t)L.H * n_frames,
                                hipMemcpyHostToDevice, s->st_up));
    HIP_TRY(c, hipEventRecord(s->ev[1], s->st_up));
    s->first_id = first_frame_id;
    return enqueue_batch(s, s->d_img, L.pitch, L.frame_stride, n_frames);
}


// ---- compressed input -------------------------------------------------------------------------------

static int ensure_decode_buffers(ffs_stream* s) {
    ffs_ctx* c = s->ctx;
    if (s->d_comp) return FFS_OK;
    const size_t es = c->pixel_bytes, nelem = (size_t)c->L.W * c->L.H;
    // bitshuffle's blocking (bshuf_default_block_size, and the loop of bshuf_blocked_wrap_fun)
    const size_t block = (size_t)kDecBlockBytes / es;
    const size_t n_full = nelem / block, rem = nelem - n_full * block;
    s->dec_block_elems = (uint32_t)block;
    s->dec_blocks = (uint32_t)(n_full + (rem >= 8 ? 1 : 0));
    s->dec_last = (uint32_t)(rem >= 8 ? rem / 8 * 8 : block);
    s->dec_tail = (uint32_t)(rem % 8);
    const size_t tab_bytes = (size_t)s->max_batch * (s->dec_blocks + 1) * sizeof(uint2);
    if (dmalloc(&s->d_comp, s->h_img_bytes + 64) != hipSuccess || dmalloc(&s->d_tab, tab_bytes) != hipSuccess
        || hipHostMalloc(reinterpret_cast<void**>(&s->h_tab), tab_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->err = "allocation of the compressed-chunk buffers failed";
        return FFS_ERR_NOMEM;
    }
    return FFS_OK;
}

static inline uint32_t be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

// Step 1 (caller's thread): validates the headers, places the chunks in the pinned staging buffer
// (unless they already are there) and starts their copy to the device.  Fills base[] = offset of every
// chunk in the staging / device buffer.
static int stage_chunks(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes, uint32_t n,
                        std::vector<size_t>& base) {
    ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    int rc = ensure_decode_buffers(s);
    if (rc != FFS_OK) return rc;
    const size_t es = c->pixel_bytes, raw_bytes = (size_t)L.W * L.H * es;
    uint32_t in_place = 0;
    for (uint32_t f = 0; f < n; ++f) {
        if (!chunks[f] || chunk_bytes[f] < 12) {
            c->err = "ffs_submit_compressed: a chunk is shorter than its 12-byte header";
            return FFS_ERR_INVALID;
        }
        const uint8_t* p = static_cast<const uint8_t*>(chunks[f]);
        uint64_t total = 0;
        for (int i = 0; i < 8; ++i) total = (total << 8) | p[i];
        if (total != raw_bytes) {
            c->err = "ffs_submit_compressed: chunk header says " + std::to_string(total) + " bytes, the context's frames have "
                     + std::to_string(raw_bytes);
            return FFS_ERR_INVALID;
        }
        if (p >= s->h_img && p + chunk_bytes[f] <= s->h_img + s->h_img_bytes) ++in_place;
    }
    if (in_place != 0 && in_place != n) {
        c->err = "ffs_submit_compressed: either all chunks lie in the stream's host buffer or none";
        return FFS_ERR_INVALID;
    }
    base.assign(n, 0);
    size_t lo = 0, hi = 0;
    if (in_place) {
        lo = SIZE_MAX;
        for (uint32_t f = 0; f < n; ++f) {
            base[f] = (size_t)(static_cast<const uint8_t*>(chunks[f]) - s->h_img);
            lo = std::min(lo, base[f]);
            hi = std::max(hi, base[f] + chunk_bytes[f]);
        }
        lo &= ~(size_t)15;
    } else {
        size_t cur = 0;
        for (uint32_t f = 0; f < n; ++f) {
            if (cur + chunk_bytes[f] > s->h_img_bytes) {
                c->err = "ffs_submit_compressed: the batch's chunks exceed the staging buffer";
                return FFS_ERR_INVALID;
            }
            std::memcpy(s->h_img + cur, chunks[f], chunk_bytes[f]);
            base[f] = cur;
            cur = (cur + chunk_bytes[f] + 15) & ~(size_t)15;
        }
        hi = cur;
    }
    if (hi > 0xFFFFFFF0ull) {
        c->err = "ffs_submit_compressed: more than 4 GiB of chunks in one batch";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipMemcpyAsync(s->d_comp + lo, s->h_img + lo, hi - lo, hipMemcpyHostToDevice, s->st_up));
    return FFS_OK;
}

// Step 2 (may run on the stream's helper thread while the chunks cross PCIe): indexes the blocks --
// each frame is a chain of [4-byte length][payload], a pointer chase of ~2 ms for 32 Eiger frames;
// the frames' chains are walked side by side so that their cache misses overlap -- and enqueues the
// table copy.  Errors go to `err`, not to the context (another thread may own that string).
static int index_blocks(ffs_stream* s, const std::vector<size_t>& base, const std::vector<size_t>& chunk_bytes,
                        std::string& err) {
    ffs_ctx* c = s->ctx;
    const uint32_t n = (uint32_t)base.size();
    const size_t es = c->pixel_bytes;
    const uint32_t nb = s->dec_blocks, stride = nb + 1;
    std::vector<size_t> pos(n, 12);
    std::atomic<bool> ok{true};
    static const bool trace = std::getenv("FFS_TRACE_SUBMIT") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto walk = [&](uint32_t f0, uint32_t f1) {  // frames [f0, f1), chains interleaved
        for (uint32_t b = 0; b < nb; ++b)
            for (uint32_t f = f0; f < f1; ++f) {
                if (pos[f] + 4 > chunk_bytes[f]) { ok = false; return; }
                const uint32_t clen = be32(s->h_img + base[f] + pos[f]);
                s->h_tab[(size_t)f * stride + b] = make_uint2((uint32_t)(base[f] + pos[f] + 4), clen);
                pos[f] += 4 + (size_t)clen;
            }
        for (uint32_t f = f0; f < f1; ++f) {
            const size_t tail = (size_t)s->dec_tail * es;
            if (pos[f] + tail > chunk_bytes[f]) ok = false;
            s->h_tab[(size_t)f * stride + nb] = make_uint2((uint32_t)(base[f] + pos[f]), (uint32_t)tail);
        }
    };
    const uint32_t n_thr = (uint64_t)n * nb >= 32768 ? std::min<uint32_t>(4, n) : 1;
    if (n_thr <= 1) {
        walk(0, n);
    } else {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < n_thr; ++t) th.emplace_back(walk, n * t / n_thr, n * (t + 1) / n_thr);
        walk(0, n / n_thr);
        for (auto& t : th) t.join();
    }
    if (trace)
        std::fprintf(stderr, "[ffs] indexed %u blocks in %.3f ms (%u threads)\n", n * nb,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), n_thr);
    if (!ok) {
        err = "ffs_submit_compressed: block lengths run past the end of a chunk";
        return FFS_ERR_INVALID;
    }
    const hipError_t e = hipMemcpyAsync(s->d_tab, s->h_tab, (size_t)n * stride * sizeof(uint2), hipMemcpyHostToDevice, s->st_up);
    if (e != hipSuccess) {
        err = std::string("hipMemcpyAsync(block table): ") + hipGetErrorString(e);
        return FFS_ERR_DEVICE;
    }
    return FFS_OK;
}

static void launch_decode(ffs_stream* s, uint32_t n, hipStream_t st) {
    ffs_ctx* c = s->ctx;
    DecodeArgs da{};
    da.comp = s->d_comp;
    da.table = s->d_tab;
    da.image = s->d_img;
    da.frame_stride = c->L.frame_stride;
    da.pitch = c->L.pitch;
    da.W = c->L.W;
    da.H = c->L.H;
    da.elem_bytes = c->pixel_bytes;
    da.blocks_per_frame = s->dec_blocks;
    da.block_elems = s->dec_block_elems;
    da.last_block_elems = s->dec_last;
    da.tail_elems = s->dec_tail;
    da.error = s->d_overflow;
    const dim3 grid(s->dec_blocks + 1, n);
    if (c->pixel_bytes == 2) hipLaunchKernelGGL(k_bshuf_lz4_decode<2>, grid, dim3(64), 0, st, da);
    else hipLaunchKernelGGL(k_bshuf_lz4_decode<4>, grid, dim3(64), 0, st, da);
}

static int ffs_submit_compressed_impl(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes,
                                     uint32_t n_frames, int64_t first_frame_id) {
    if (!s || !chunks || !chunk_bytes) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream already has a batch in flight: call ffs_wait() first";
        return FFS_ERR_INVALID;
    }
    if (n_frames == 0 || n_frames > s->max_batch) {
        c->err = "n_frames must be in 1..max_batch";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    s->dev_input = false;
    HIP_TRY(c, hipEventRecord(s->ev[0], s->st_up));
    std::vector<size_t> base;
    int rc = stage_chunks(s, chunks, chunk_bytes, n_frames, base);
    if (rc != FFS_OK) return rc;
    // The rest -- block index, table copy, decode kernel and the hot path's launches -- is enqueued by a
    // helper thread, so that the caller gets its thread back while the index is built; ffs_wait joins it.
    s->first_id = first_frame_id;
    s->n_frames = n_frames;
    mark_busy(s);
    s->job_rc = FFS_OK;
    s->job_err.clear();
    const ffs_params snap = c->params;
    std::vector<size_t> sizes(chunk_bytes, chunk_bytes + n_frames);
    s->job = std::thread([s, c, snap, n_frames, base = std::move(base), sizes = std::move(sizes)]() {
        if (hipSetDevice(c->device) != hipSuccess) {
            s->job_rc = FFS_ERR_DEVICE;
            s->job_err = "hipSetDevice failed on the stream's helper thread";
            return;
        }
        int r = index_blocks(s, base, sizes, s->job_err);
        if (r == FFS_OK) {
            (void)hipGetLastError();
            // the decode kernel runs with the dense kernels (in their order), behind the copies of its input
            hipError_t e = hipSuccess;
            hipStream_t dst = c->knobs.decode_dense ? s->st : s->st_up;
            if (dst != s->st_up) {
                e = hipEventRecord(s->ev[6], s->st_up);
                if (e == hipSuccess) e = hipStreamWaitEvent(dst, s->ev[6], 0);
            }
            launch_decode(s, n_frames, dst);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(s->ev[1], dst);
            if (e != hipSuccess) {
                s->job_err = std::string("decode launch: ") + hipGetErrorString(e);
                r = FFS_ERR_DEVICE;
            }
        }
        if (r == FFS_OK) {
            r = enqueue_batch(s, s->d_img, c->L.pitch, c->L.frame_stride, n_frames, &snap);
            if (r != FFS_OK) s->job_err = c->err;
        }
        s->job_rc = r;
    });
    return FFS_OK;
}

extern "C" int ffs_decode_only(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes, uint32_t n_frames,
                               uint32_t iters, float* ms_decode, void* host_out) {
    if (!s || !chunks || !chunk_bytes || iters == 0) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy || n_frames == 0 || n_frames > s->max_batch) {
        c->err = "ffs_decode_only: stream busy or n_frames out of range";
        return FFS_ERR_INVALID;
    }
    const Layout& L = c->L;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<size_t> base;
    int rc = stage_chunks(s, chunks, chunk_bytes, n_frames, base);
    if (rc == FFS_OK) {
        std::string err;
        rc = index_blocks(s, base, std::vector<size_t>(chunk_bytes, chunk_bytes + n_frames), err);
        if (rc != FFS_OK) c->err = err;
    }
    if (rc != FFS_OK) {
        (void)hipStreamSynchronize(s->st_up);
        return rc;
    }
    (void)hipGetLastError();
    HIP_TRY(c, hipEventRecord(s->ev[0], s->st_up));
    for (uint32_t i = 0; i < iters; ++i) launch_decode(s, n_frames, s->st_up);
    HIP_TRY(c, hipEventRecord(s->ev[1], s->st_up));
    HIP_TRY(c, hipGetLastError());
    uint32_t flag = 0;
    HIP_TRY(c, hipMemcpyAsync(&flag, s->d_overflow, 4, hipMemcpyDeviceToHost, s->st_up));
    HIP_TRY(c, hipStreamSynchronize(s->st_up));
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
    if (ms_decode) *ms_decode = ms / iters;
    if (host_out) {
        const size_t row = (size_t)L.W * c->pixel_bytes;
        HIP_TRY(c, hipMemcpy2D(host_out, row, s->d_img, L.pitch, row, (size_t)L.H * n_frames, hipMemcpyDeviceToHost));
    }
    if (flag & 4u) {
        HIP_TRY(c, hipMemset(s->d_overflow, 0, 4));
        c->err = "corrupt bitshuffle-LZ4 chunk: an LZ4 block did not decode to its block size";
        return FFS_ERR_INVALID;
    }
    return FFS_OK;
}

static int ensure_list_host(ffs_stream* s) {
    ffs_ctx* c = s->ctx;
    if (!s->h_list_k) {
        const size_t bytes = (size_t)s->max_batch * s->cap * 4;
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_list_k), bytes, hipHostMallocDefault));
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_list_i), bytes, hipHostMallocDefault));
    }
    return FFS_OK;
}

static int ffs_wait_impl(ffs_stream* s, const ffs_frame_result** results, uint32_t* n_results) {
    if (!s) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (!s->busy) {
        c->err = "ffs_wait: nothing submitted";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (s->job.joinable()) {
        s->job.join();
        if (s->job_rc != FFS_OK) {
            (void)hipStreamSynchronize(s->st_up);
            mark_idle(s);
            c->err = s->job_err;
            return s->job_rc;
        }
    }
    HIP_TRY(c, hipEventSynchronize(s->ev[4]));
    const uint32_t n = s->n_frames;
    const size_t B = s->max_batch;
    const Layout& L = c->L;
    const ffs_params& p = s->batch_params;
    const uint32_t* h_ns = s->h_counts;
    const uint32_t* h_nc = s->h_counts + B;
    const uint32_t* h_sm = s->h_counts + 2 * B;
    uint32_t overflow = s->h_counts[10 * B];
    if (s->chain_mode) {  // k_frame_chain: one flag word per frame
        overflow = 0;
        for (uint32_t f = 0; f < n; ++f) overflow |= s->h_counts[10 * B + 1 + f];
    }
    mark_idle(s);
    s->ovf.clear();
    if (overflow) {
        s->bits_dirty = true;
        (void)hipMemsetAsync(s->d_overflow, 0, 4, s->st2);
        (void)hipStreamSynchronize(s->st2);
        if (overflow & 4u) {
            c->err = "corrupt bitshuffle-LZ4 chunk: an LZ4 block did not decode to its block size";
            return FFS_ERR_INVALID;
        }
        if (overflow & 8u) {
            // more bright-window pixels than the list k_stream_u16 hands to k_bright_fix holds (a batch of
            // saturated frames): run the batch again through the two-kernel threshold path, which has no such list
            s->force_variant = 1;
            int rc = enqueue_batch(s, s->cur_img, s->cur_pitch, s->cur_fstride, s->n_frames, &s->batch_params);
            s->force_variant = -1;
            if (rc != FFS_OK) return rc;
            return ffs_wait_impl(s, results, n_results);
        }
        // A frame with more strong pixels than the stream's lists hold (flag 1) or more components than its
        // record buffers (flag 2) -- an ice ring, the direct beam.  The reference has no such limit (std::map of
        // signals, connected_components.cc:24-32), so neither may the drop-in: the other frames of the batch are
        // complete (lists and records are per frame), and each frame that did not fit is run again on its own on
        // a one-frame stream with room for it (kept for the next time).
        for (uint32_t f = 0; f < n; ++f) {
            if (h_ns[f] <= s->cap && h_nc[f] <= s->max_comp) continue;
            uint64_t need_cap = std::max<uint64_t>(h_ns[f], s->cap);
            uint64_t need_comp = h_ns[f] > s->cap ? need_cap : std::max<uint64_t>(h_nc[f], s->max_comp);  // list truncated: count unknown
            for (int attempt = 0;; ++attempt) {
                if (s->big && (s->big->cap < need_cap || s->big->max_comp < need_comp)) {
                    ffs_stream_destroy(s->big);
                    s->big = nullptr;
                }
                if (!s->big) {
                    const uint64_t npx = (uint64_t)L.W * L.H;
                    const uint32_t bc = (uint32_t)std::min<uint64_t>(npx, need_cap + need_cap / 4 + 1024);
                    const uint32_t bm = (uint32_t)std::min<uint64_t>(bc, need_comp + need_comp / 4 + 1024);
                    int rc = stream_create_sized(c, 1, bc, bm, &s->big);
                    if (rc != FFS_OK) return rc;  // a real out-of-memory
                }
                ffs_stream* b = s->big;
                const uint8_t* img = static_cast<const uint8_t*>(s->cur_img) + (size_t)f * s->cur_fstride;
                b->first_id = s->first_id + f;
                b->dev_input = true;
                b->ev1_pending = true;
                int rc = enqueue_batch(b, img, s->cur_pitch, s->cur_fstride, 1, &s->batch_params);
                if (rc != FFS_OK) return rc;
                HIP_TRY(c, hipEventSynchronize(b->ev[4]));
                const uint32_t b_ovf = b->chain_mode ? b->h_counts[10 * (size_t)b->max_batch + 1] : b->h_counts[10 * (size_t)b->max_batch];
                if (b_ovf & 3u) {  // only the component count can still be short (it was a guess while the list was cut)
                    (void)hipMemsetAsync(b->d_overflow, 0, 4, b->st2);
                    (void)hipStreamSynchronize(b->st2);
                    mark_idle(b);
                    b->bits_dirty = true;
                    need_cap = std::max<uint64_t>(need_cap, b->h_counts[0]);
                    need_comp = std::max<uint64_t>(need_comp * 2, b->h_counts[b->max_batch]);
                    if (attempt >= 4) {
                        c->err = "a frame still overflows its one-frame stream";
                        return FFS_ERR_OVERFLOW;
                    }
                    continue;
                }
                const ffs_frame_result* br = nullptr;
                uint32_t bn = 0;
                rc = ffs_wait_impl(b, &br, &bn);
                if (rc != FFS_OK) return rc;
                s->ovf.emplace_back();
                OverflowFrame& o = s->ovf.back();
                o.frame = f;
                o.res = br[0];
                o.boxes.assign(br[0].boxes, br[0].boxes + br[0].n_boxes);
                if (br[0].reflections) o.refls.assign(br[0].reflections, br[0].reflections + br[0].n_reflections);
                if (br[0].strong_k) {
                    o.k.assign(br[0].strong_k, br[0].strong_k + br[0].num_strong_pixels);
                    o.inten.assign(br[0].strong_intensity, br[0].strong_intensity + br[0].num_strong_pixels);
                }
                break;
            }
        }
    }
    auto overflow_frame = [&](uint32_t f) -> const OverflowFrame* {
        for (const OverflowFrame& o : s->ovf)
            if (o.frame == f) return &o;
        return nullptr;
    };
    uint64_t total_recs = 0;
    uint32_t max_ns = 0;
    for (uint32_t f = 0; f < n; ++f) {
        total_recs += std::min<uint32_t>(h_nc[f], s->max_comp);  // (the kernels never write more than max_comp per frame)
        max_ns = std::max(max_ns, std::min<uint32_t>(h_ns[f], s->cap));
    }
    bool second_phase = false;
    if (total_recs > s->spec_recs_copied) {  // more records than the speculative copy brought: fetch the rest
        const size_t rb = s->wire2 ? sizeof(WireRec2) : sizeof(ReflOut);
        HIP_TRY(c, hipMemcpyAsync(reinterpret_cast<uint8_t*>(s->h_recs) + s->spec_recs_copied * rb,
                                  reinterpret_cast<const uint8_t*>(s->d_recs) + s->spec_recs_copied * rb,
                                  (total_recs - s->spec_recs_copied) * rb, hipMemcpyDeviceToHost, s->st2));
        second_phase = true;
    }
    s->spec_recs_per_frame = std::max<uint32_t>(s->spec_recs_per_frame,
                                                (uint32_t)std::min<uint64_t>(s->max_comp, (total_recs / n + 1) * 5 / 4));
    if (p.want_strong_list && max_ns) {
        second_phase = true;
        int rc = ensure_list_host(s);
        if (rc != FFS_OK) return rc;
        HIP_TRY(c, hipMemcpy2DAsync(s->h_list_k, (size_t)s->cap * 4, s->d_list_k, (size_t)s->cap * 4,
                                    (size_t)max_ns * 4, n, hipMemcpyDeviceToHost, s->st2));
        HIP_TRY(c, hipMemcpy2DAsync(s->h_list_i, (size_t)s->cap * 4, s->d_list_i, (size_t)s->cap * 4,
                                    (size_t)max_ns * 4, n, hipMemcpyDeviceToHost, s->st2));
    }
    if (p.want_strong_mask) {
        second_phase = true;
        if (!s->h_mask)
            HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_mask), B * (size_t)L.W * L.H, hipHostMallocDefault));
        // the reference's full-mask D2H (spotfinder.cc:887-894), all frames of the batch in one 2D copy
        HIP_TRY(c, hipMemcpy2DAsync(s->h_mask, L.W, s->d_sbytes, L.bpitch, L.W, (size_t)L.H * n,
                                    hipMemcpyDeviceToHost, s->st2));
    }
    hipEvent_t last = s->ev[4];
    if (second_phase) {
        HIP_TRY(c, hipEventRecord(s->ev[5], s->st2));
        HIP_TRY(c, hipEventSynchronize(s->ev[5]));
        last = s->ev[5];
    }
    s->timings[0] = 0.0f;
    if (!s->dev_input) (void)hipEventElapsedTime(&s->timings[0], s->ev[0], s->ev[1]);
    (void)hipEventElapsedTime(&s->timings[1], s->ev[1], s->ev[2]);
    (void)hipEventElapsedTime(&s->timings[2], s->ev[2], s->ev3_is_ev4 ? s->ev[4] : s->ev[3]);
    (void)hipEventElapsedTime(&s->timings[3], s->ev3_is_ev4 ? s->ev[4] : s->ev[3], last);
    (void)hipEventElapsedTime(&s->timings[4], s->dev_input ? s->ev[1] : s->ev[0], last);

    // assemble: boxes = components surviving the min-size filter (connected_components.cc:122-135),
    // reflections = components surviving filter_reflections (:207-236); both keep label order.
    s->results.assign(n, ffs_frame_result{});
    s->boxes.clear();
    s->refls.clear();
    s->boxes.reserve(total_recs);
    if (p.want_reflections) s->refls.reserve(total_recs);
    std::vector<size_t> box_at(n), refl_at(n);
    const ReflOut* rec = s->h_recs;
    const WireRec2* wrec = reinterpret_cast<const WireRec2*>(s->h_recs);
    for (uint32_t f = 0; f < n; ++f) {
        box_at[f] = s->boxes.size();
        refl_at[f] = s->refls.size();
        const uint32_t nc = std::min<uint32_t>(h_nc[f], s->max_comp);
        if (s->chain_mode) wrec = reinterpret_cast<const WireRec2*>(s->h_recs) + (size_t)f * s->max_comp;  // k_frame_chain: every frame has its own record area
        if (const OverflowFrame* o = overflow_frame(f)) {  // re-run on the one-frame stream: skip the cut records
            if (s->wire2) wrec += nc;
            else rec += nc;
            s->boxes.insert(s->boxes.end(), o->boxes.begin(), o->boxes.end());
            if (p.want_reflections) s->refls.insert(s->refls.end(), o->refls.begin(), o->refls.end());
            continue;
        }
        if (s->wire2) {
            for (uint32_t q = 0; q < nc; ++q, ++wrec) {
                const uint32_t npx = wrec->npx_flags & 0x3FFFFFFFu, flags = wrec->npx_flags >> 30;
                if (p.min_spot_size == 0 || npx >= p.min_spot_size)
                    s->boxes.push_back(ffs_box{wrec->x_min, wrec->y_min, wrec->x_max, wrec->y_max, (int32_t)npx});
                if (p.want_reflections && flags == 0) {
                    ffs_reflection r{};
                    r.x_min = wrec->x_min; r.x_max = wrec->x_max; r.y_min = wrec->y_min; r.y_max = wrec->y_max;
                    r.z_min = 0; r.z_max = 0;
                    r.num_pixels = (int32_t)npx;
                    r.com_x = wrec->com_x; r.com_y = wrec->com_y; r.com_z = 0.5f;  // z = 0 for a single frame
                    r.peak_x = wrec->peak_x; r.peak_y = wrec->peak_y; r.peak_z = 0;
                    r.peak_intensity = wrec->peak_intensity;
                    r.peak_centroid_distance = wrec->peak_centroid_distance;
                    r.flags = 0;
                    r.sum_intensity = wrec->sum_intensity;
                    s->refls.push_back(r);
                }
            }
            continue;
        }
        for (uint32_t q = 0; q < nc; ++q, ++rec) {
            if (p.min_spot_size == 0 || (uint32_t)rec->num_pixels >= p.min_spot_size)
                s->boxes.push_back(ffs_box{rec->x_min, rec->y_min, rec->x_max, rec->y_max, rec->num_pixels});
            if (p.want_reflections && rec->flags == 0) {
                ffs_reflection r;
                std::memcpy(&r, rec, sizeof(r));
                s->refls.push_back(r);
            }
        }
    }
    for (uint32_t f = 0; f < n; ++f) {
        ffs_frame_result& r = s->results[f];
        const uint32_t* sm = h_sm + (size_t)f * 8;
        r.frame_id = s->first_id + f;
        r.num_strong_pixels = h_ns[f];
        r.num_strong_pixels_filtered = sm[1];
        r.n_components = h_nc[f];
        r.n_boxes = sm[0];
        r.boxes = s->boxes.data() + box_at[f];
        r.n_reflections = p.want_reflections ? sm[2] : 0;
        r.reflections = p.want_reflections ? s->refls.data() + refl_at[f] : nullptr;
        r.n_filtered_size = sm[3];
        r.n_filtered_sep = sm[4];
        if (p.want_strong_list) {
            r.strong_k = s->h_list_k ? s->h_list_k + (size_t)f * s->cap : nullptr;
            r.strong_intensity = s->h_list_i ? s->h_list_i + (size_t)f * s->cap : nullptr;
        }
        if (p.want_strong_mask) r.strong_mask = s->h_mask + (size_t)f * L.W * L.H;
        if (const OverflowFrame* o = overflow_frame(f)) {
            const ffs_frame_result& b = o->res;
            r.num_strong_pixels = b.num_strong_pixels;
            r.num_strong_pixels_filtered = b.num_strong_pixels_filtered;
            r.n_components = b.n_components;
            r.n_boxes = b.n_boxes;
            r.n_reflections = p.want_reflections ? b.n_reflections : 0;
            r.n_filtered_size = b.n_filtered_size;
            r.n_filtered_sep = b.n_filtered_sep;
            if (p.want_strong_list) {
                r.strong_k = o->k.data();
                r.strong_intensity = o->inten.data();
            }
        }
    }
    if (results) *results = s->results.data();
    if (n_results) *n_results = n;
    return FFS_OK;
}

extern "C" int ffs_stream_batch_arrays(ffs_stream* s, const ffs_box** boxes, uint32_t* n_boxes,
                                       const ffs_reflection** refls, uint32_t* n_refls) {
    if (!s) return FFS_ERR_INVALID;
    if (boxes) *boxes = s->boxes.data();
    if (n_boxes) *n_boxes = (uint32_t)s->boxes.size();
    if (refls) *refls = s->refls.data();
    if (n_refls) *n_refls = (uint32_t)s->refls.size();
    return FFS_OK;
}

extern "C" int ffs_stream_timings(ffs_stream* s, float ms[5]) {
    if (!s || !ms) return FFS_ERR_INVALID;
    std::memcpy(ms, s->timings, sizeof(s->timings));
    return FFS_OK;
}

extern "C" int ffs_stream_spot_centres(ffs_stream* s, float* rows4, uint32_t cap, uint32_t* n_written) {
    if (!s || !rows4) return FFS_ERR_INVALID;
    if (s->busy) {
        s->ctx->err = "ffs_stream_spot_centres: a batch is in flight";
        return FFS_ERR_INVALID;
    }
    uint32_t n = 0;
    uint64_t wanted = 0;
    for (const ffs_frame_result& r : s->results) {
        // the id's low 32 bits as a bit pattern: as a float VALUE ids would collide from 2^24 on
        const uint32_t id_bits = (uint32_t)((uint64_t)r.frame_id & 0xFFFFFFFFull);
        float id;
        std::memcpy(&id, &id_bits, 4);
        wanted += r.n_reflections;
        for (uint32_t q = 0; q < r.n_reflections && n < cap; ++q, ++n) {
            float* row = rows4 + (size_t)n * 4;
            row[0] = id;
            row[1] = r.reflections[q].com_x;
            row[2] = r.reflections[q].com_y;
            row[3] = r.reflections[q].com_z;
        }
    }
    // last row: (rows written, rows wanted) as uint32 bit patterns -- wanted > written tells the receiver
    // that `cap` was too small (nothing is dropped silently)
    float* last = rows4 + (size_t)cap * 4;
    const uint32_t tail[4] = {n, (uint32_t)std::min<uint64_t>(wanted, 0xFFFFFFFFull), 0u, 0u};
    std::memcpy(last, tail, sizeof(tail));
    if (n_written) *n_written = n;
    return wanted > n ? FFS_ERR_OVERFLOW : FFS_OK;
}

extern "C" int ffs_stream_debug_planes(ffs_stream* s, const uint8_t** strong_bytes, size_t* mask_pitch,
                                       size_t* mask_fstride) {
    if (!s) return FFS_ERR_INVALID;
    if (strong_bytes) *strong_bytes = s->d_sbytes;
    if (mask_pitch) *mask_pitch = s->ctx->L.bpitch;
    if (mask_fstride) *mask_fstride = s->ctx->L.bytes_frame_stride;
    return FFS_OK;
}

extern "C" int ffs_stream_debug_bitplane(ffs_stream* s, uint32_t frame, int which, uint8_t* host_out) {
    if (!s || !host_out || which < 0 || which > 2) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    if (s->busy || frame >= s->max_batch) {
        c->err = "ffs_stream_debug_bitplane: stream busy or frame out of range";
        return FFS_ERR_INVALID;
    }
    const uint8_t* src = which == 0 ? s->d_bits : which == 1 ? s->d_dplane : s->d_eplane;
    if (!src) {
        c->err = "ffs_stream_debug_bitplane: that plane exists only after an extended-dispersion batch";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (which == 0 && s->bits_cleared && !s->dense_valid) {
        // the compaction consumed (and cleared) the plane and nobody asked for the byte mask: the strong-pixel list has them
        const uint32_t ns = std::min<uint32_t>(s->h_counts[frame], s->cap);
        std::vector<uint32_t> ks(ns);
        if (ns) HIP_TRY(c, hipMemcpy(ks.data(), s->d_list_k + (size_t)frame * s->cap, (size_t)ns * 4, hipMemcpyDeviceToHost));
        std::memset(host_out, 0, (size_t)L.W * L.H);
        for (uint32_t k : ks) host_out[k] = 1;
        return FFS_OK;
    }
    if (which == 0 && s->bits_cleared) {
        // the compaction consumed (and cleared) the plane; the byte mask holds the same pixels
        HIP_TRY(c, hipMemcpy2D(host_out, L.W, s->d_sbytes + (size_t)frame * L.bytes_frame_stride, L.bpitch, L.W, L.H,
                               hipMemcpyDeviceToHost));
        return FFS_OK;
    }
    std::vector<uint8_t> packed(L.plane_frame_stride);
    HIP_TRY(c, hipMemcpy(packed.data(), src + (size_t)frame * L.plane_frame_stride, packed.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.H; ++y)
        for (int x = 0; x < L.W; ++x)
            host_out[(size_t)y * L.W + x] = (packed[(size_t)y * L.mpitch + (x >> 3)] >> (x & 7)) & 1u;
    return FFS_OK;
}

extern "C" int ffs_bench_threshold(ffs_stream* s, const void* device_pixels, size_t pitch, size_t fstride,
                                   uint32_t n_frames, uint32_t iters, float* ms_candidate, float* ms_exact) {
    if (!s || !device_pixels || iters == 0) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream busy";
        return FFS_ERR_INVALID;
    }
    int rc = check_layout(s, pitch, fstride, n_frames);
    if (rc != FFS_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    s->batch_params = c->params;
    const bool ext = c->params.algorithm == FFS_ALGO_DISPERSION_EXTENDED;
    if (ext) {
        rc = ensure_extended_buffers(s);
        if (rc != FFS_OK) return rc;
    }
    const ThresholdArgs ta = make_threshold_args(s, device_pixels, pitch, fstride, n_frames);
    // (1) `iters` launches of the dense kernel (candidates / extended first pass) back to back, HIP
    //     events on this stream
    HIP_TRY(c, hipEventRecord(s->ev[0], s->st));
    for (uint32_t i = 0; i < iters; ++i) ext ? launch_ext_first(s, ta, n_frames) : launch_candidates(s, ta, n_frames);
    HIP_TRY(c, hipEventRecord(s->ev[1], s->st));
    // (2) dense + sparse stages together (the exact kernel filters the plane in place, so it must
    //     always be preceded by the candidate kernel); sparse = (pair - dense)
    for (uint32_t i = 0; i < iters; ++i) {
        if (ext) {
            launch_extended(s, ta, n_frames);
        } else {
            launch_candidates(s, ta, n_frames);
            launch_exact(s, ta, n_frames);
        }
    }
    HIP_TRY(c, hipEventRecord(s->ev[2], s->st));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventSynchronize(s->ev[2]));
    float t1 = 0, t2 = 0;
    HIP_TRY(c, hipEventElapsedTime(&t1, s->ev[0], s->ev[1]));
    HIP_TRY(c, hipEventElapsedTime(&t2, s->ev[1], s->ev[2]));
    if (ms_candidate) *ms_candidate = t1 / iters;
    if (ms_exact) *ms_exact = std::max(0.0f, (t2 - t1) / iters);
    s->bits_dirty = true;  // no compaction ran: the strong plane still holds this batch's bits
    s->counts_dirty = true;
    return FFS_OK;
}

// ---- measured memory ceiling (a ceiling measured on the box beside the nominal 8 TB/s) ----
// k_probe<0>: every byte of the batch's pixel buffer is read once (16 B per lane, consecutive).
// k_probe<1|2>: the same reads plus one 8-byte zero store per 16 bytes read into the byte-mask buffer -- the 2:1
// read/write mix of the threshold kernel (2 B pixel in, 1 B mask out), again perfectly linear.
template <int WRITE>  // 0 = reads only, 1 = plain stores, 2 = non-temporal stores
__global__ __launch_bounds__(256) void k_probe(const uint4* src, uint2* dst, uint64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (WRITE == 1) dst[i] = make_uint2(0u, 0u);
        if (WRITE == 2) {
            __builtin_nontemporal_store(0u, &dst[i].x);
            __builtin_nontemporal_store(0u, &dst[i].y);
        }
    }
    if (acc == 0x9E3779B9u) *sink = acc;  // keeps the loads alive
}

extern "C" int ffs_bench_hbm(ffs_stream* s, uint32_t iters, float* read_gbps, float* mix_gbps) {
    if (!s || iters == 0) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream busy";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t n16 = (uint64_t)s->max_batch * c->L.frame_stride / 16;
    // the byte-mask buffer holds half as many bytes as the pixel buffer for 16-bit pixels, a quarter for 32-bit
    const uint64_t n16w = std::min<uint64_t>(n16, (uint64_t)s->max_batch * c->L.bytes_frame_stride / 8);
    const uint4* src = reinterpret_cast<const uint4*>(s->d_img);
    uint2* dst = reinterpret_cast<uint2*>(s->d_sbytes);
    float out[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
        const uint64_t n = mode ? n16w : n16;
        auto launch = [&]() {
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(4096), dim3(256), 0, s->st, src, dst, n, s->d_tile_counts);
            else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(4096), dim3(256), 0, s->st, src, dst, n, s->d_tile_counts);
            else hipLaunchKernelGGL(k_probe<0>, dim3(4096), dim3(256), 0, s->st, src, dst, n, s->d_tile_counts);
        };
        launch();  // warm-up
        HIP_TRY(c, hipEventRecord(s->ev[0], s->st));
        for (uint32_t i = 0; i < iters; ++i) launch();
        HIP_TRY(c, hipEventRecord(s->ev[1], s->st));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventSynchronize(s->ev[1]));
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
        const double bytes = (double)n * 16.0 + (mode ? (double)n * 8.0 : 0.0);
        out[mode] = (float)(bytes * iters / (ms * 1e-3) / 1e9);
    }
    if (read_gbps) *read_gbps = out[0];
    if (mix_gbps) *mix_gbps = std::max(out[1], out[2]);  // the better of plain and non-temporal stores
    return FFS_OK;
}

__global__ void k_selftest_sqrt(unsigned long long begin, unsigned long long end, unsigned long long* out) {
    unsigned long long acc = 0;
    for (unsigned long long n = begin + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; n < end;
         n += (unsigned long long)gridDim.x * blockDim.x)
        acc += (unsigned long long)__double_as_longlong(__builtin_sqrt((double)n));
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

extern "C" int ffs_selftest_sqrt(ffs_ctx* c, uint64_t begin, uint64_t end, uint64_t* sum_of_bits) {
    if (!c || !sum_of_bits || end < begin) return FFS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&d), 8));
    HIP_TRY(c, hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(2048), dim3(256), 0, 0, begin, end, d);
    hipError_t e = hipMemcpy(sum_of_bits, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(c, e);
    return FFS_OK;
}

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

extern "C" int ffs_stack3d_create(ffs_ctx* c, uint64_t max_total, ffs_stack3d** out) {
    if (!c || !out) return FFS_ERR_INVALID;
    *out = nullptr;
    return stack3d_create_impl(c, max_total, out);
}

extern "C" void ffs_stack3d_destroy(ffs_stack3d* st) {
    if (!st) return;
    ffs_ctx* c = st->ctx;
    (void)hipSetDevice(c->device);
    {
        std::lock_guard<std::mutex> lock(c->stream_mu);
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

static void stack3d_free(ffs_stack3d* st) {
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
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static std::mutex g_multi_mu;
static RcclApi g_rccl;
static std::vector<int> g_comm_devices;     // distinct devices, rank = position
static std::vector<ncclComm_t> g_comms;     // one communicator per rank (ncclCommInitAll)
static std::string g_multi_transport = "none";
static std::string g_gather_want = "rccl";  // FFS_GATHER / the transport argument, as given to ffs_multi_init
static bool g_gather_forced = false;          // ... explicitly (then RCCL is used even between contexts on one GPU)

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
    const std::string want = transport ? transport : (std::getenv("FFS_GATHER") ? std::getenv("FFS_GATHER") : "rccl");
    g_gather_forced = transport != nullptr || std::getenv("FFS_GATHER") != nullptr;
    g_gather_want = want;
    if (!g_comms.empty() && distinct == g_comm_devices) return FFS_OK;
    if (!g_comms.empty() && g_rccl.CommDestroy) {
        for (ncclComm_t cm : g_comms) (void)g_rccl.CommDestroy(cm);
        g_comms.clear();
    }
    g_comm_devices = distinct;
    g_multi_transport = distinct.size() > 1 ? "peer" : "none";
    if (want != "rccl") return FFS_OK;
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
    g_multi_transport = "rccl";
    return FFS_OK;
}

extern "C" const char* ffs_multi_transport(void) {
    std::lock_guard<std::mutex> lock(g_multi_mu);
    return g_multi_transport.c_str();
}

// Lists of a batch processed on ANOTHER context (same detector geometry, usually another GPU) into this stack:
// packed end to end on the source device, then one transfer per array.
static int stack3d_add_batch_remote(ffs_stack3d* st, ffs_stream* s, uint64_t more, uint32_t biggest) {
    ffs_ctx* c = st->ctx;       // home
    ffs_ctx* sc = s->ctx;       // source
    const uint32_t nf = s->n_frames;
    // source side: pack buffers and table (offsets from 0), on the source stream
    HIP_TRY(c, hipSetDevice(sc->device));
    if (!s->d_pack_k) {
        const size_t cap_all = (size_t)s->max_batch * s->cap;
        if (dmalloc(&s->d_pack_k, cap_all * 4) != hipSuccess || dmalloc(&s->d_pack_i, cap_all * 4) != hipSuccess
            || dmalloc(&s->d_pack_tab, (size_t)s->max_batch * sizeof(StackSlice)) != hipSuccess
            || hipHostMalloc(reinterpret_cast<void**>(&s->h_pack_tab), (size_t)s->max_batch * sizeof(StackSlice), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->err = "allocation of the list pack buffers failed";
            return FFS_ERR_NOMEM;
        }
    }
    uint32_t at = 0;
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t n = s->results[f].num_strong_pixels;
        s->h_pack_tab[f] = StackSlice{0u, at, n, 0u};
        at += n;
    }
    if (biggest) {
        STK_TRY(c, hipMemcpyAsync(s->d_pack_tab, s->h_pack_tab, (size_t)nf * sizeof(StackSlice), hipMemcpyHostToDevice, s->st2));
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_stack_append, dim3(std::min<uint32_t>(64, (biggest + 255) / 256), nf), dim3(256), 0, s->st2,
                           s->d_list_k, s->d_list_i, (uint64_t)s->cap, s->d_pack_tab, s->d_pack_k, s->d_pack_i);
        STK_TRY(c, hipGetLastError());
    }
    uint32_t* dst_k = st->a_k.p + st->arrived;
    uint32_t* dst_i = st->a_i.p + st->arrived;
    std::lock_guard<std::mutex> lock(g_multi_mu);
    const int r_src = comm_rank_of(sc->device), r_home = comm_rank_of(c->device);
    const bool use_rccl = !g_comms.empty() && r_src >= 0 && r_home >= 0 && g_gather_want == "rccl"
                          && (sc->device != c->device || g_gather_forced);
    if (more == 0) {
        STK_TRY(c, hipStreamSynchronize(s->st2));
        return FFS_OK;
    }
    if (use_rccl) {
        // one group: the source rank sends on its stream, the home rank receives on the stack's stream
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.Send(s->d_pack_k, more, ncclUint32, r_home, g_comms[r_src], s->st2);
        if (r == ncclSuccess) r = g_rccl.Recv(dst_k, more, ncclUint32, r_src, g_comms[r_home], st->st);
        if (r == ncclSuccess) r = g_rccl.Send(s->d_pack_i, more, ncclUint32, r_home, g_comms[r_src], s->st2);
        if (r == ncclSuccess) r = g_rccl.Recv(dst_i, more, ncclUint32, r_src, g_comms[r_home], st->st);
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = re;
        if (r != ncclSuccess) {
            c->err = std::string("RCCL send/recv of the strong-pixel lists: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
            return FFS_ERR_DEVICE;
        }
        STK_TRY(c, hipStreamSynchronize(s->st2));
        HIP_TRY(c, hipSetDevice(c->device));
        STK_TRY(c, hipStreamSynchronize(st->st));
        return FFS_OK;
    }
    // peer copy (or plain device-to-device when both contexts sit on one GPU): after the pack has finished
    STK_TRY(c, hipStreamSynchronize(s->st2));
    HIP_TRY(c, hipSetDevice(c->device));
    if (sc->device == c->device) {
        STK_TRY(c, hipMemcpyAsync(dst_k, s->d_pack_k, more * 4, hipMemcpyDeviceToDevice, st->st));
        STK_TRY(c, hipMemcpyAsync(dst_i, s->d_pack_i, more * 4, hipMemcpyDeviceToDevice, st->st));
    } else {
        STK_TRY(c, hipMemcpyPeerAsync(dst_k, c->device, s->d_pack_k, sc->device, more * 4, st->st));
        STK_TRY(c, hipMemcpyPeerAsync(dst_i, c->device, s->d_pack_i, sc->device, more * 4, st->st));
    }
    STK_TRY(c, hipStreamSynchronize(st->st));
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
    if (s->ctx != st->ctx) {
        // a batch from another context: same detector, another GPU (or another context on this one)
        ffs_ctx* home = st->ctx;
        if (home->L.W != c->L.W || home->L.H != c->L.H) {
            c->err = "ffs_stack3d_add_batch: the stream's context has another frame shape than the stack's";
            return FFS_ERR_INVALID;
        }
        if (!s->ovf.empty()) {
            c->err = "ffs_stack3d_add_batch: a frame overflowed the lists of a stream on another device; use a larger max_strong_per_frame";
            return FFS_ERR_OVERFLOW;
        }
        HIP_TRY(home, hipSetDevice(home->device));
        int rc = stack3d_reserve(st, more);
        if (rc != FFS_OK) { c->err = home->err; return rc; }
        uint32_t big = 0;
        for (uint32_t f = 0; f < nf; ++f) big = std::max(big, s->results[f].num_strong_pixels);
        rc = stack3d_add_batch_remote(st, s, more, big);
        if (rc != FFS_OK) { c->err = home->err; return rc; }
        uint64_t at = st->arrived;
        for (uint32_t f = 0; f < nf; ++f) {
            const uint32_t n = s->results[f].num_strong_pixels;
            st->slices[s->results[f].frame_id] = ffs_stack3d::Slice{(uint32_t)at, n};
            at += n;
        }
        st->arrived = at;
        (void)hipSetDevice(c->device);
        return FFS_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = stack3d_reserve(st, more);
    if (rc != FFS_OK) return rc;
    // The append runs in the stream's own sparse stream: behind the launch that wrote the lists, ahead of the one that
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
    uint64_t at = st->arrived;
    uint32_t biggest = 0;
    bool host_lists = false;
    for (uint32_t f = 0; f < nf; ++f) {
        const ffs_frame_result& r = s->results[f];
        const OverflowFrame* o = nullptr;
        for (const OverflowFrame& q : s->ovf)
            if (q.frame == f) o = &q;
        const uint32_t n = r.num_strong_pixels;
        // a frame that did not fit the stream's lists was re-run on its one-frame stream: its list is on the host
        h_tab[f] = StackSlice{0u, (uint32_t)at, o ? 0u : n, 0u};
        if (o && n) {
            if (o->k.size() != n) {  // (lists are only kept with want_strong_list)
                c->err = "ffs_stack3d_add_batch: a frame overflowed the stream's lists; set want_strong_list (or a larger "
                         "max_strong_per_frame) for rotation sweeps";
                return FFS_ERR_OVERFLOW;
            }
            STK_TRY(c, hipMemcpyAsync(st->a_k.p + at, o->k.data(), (size_t)n * 4, hipMemcpyHostToDevice, s->st2));
            STK_TRY(c, hipMemcpyAsync(st->a_i.p + at, o->inten.data(), (size_t)n * 4, hipMemcpyHostToDevice, s->st2));
            host_lists = true;
        }
        if (!o) biggest = std::max(biggest, n);
        st->slices[r.frame_id] = ffs_stack3d::Slice{(uint32_t)at, n};
        at += n;
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

// ---- guarded entry points (see `guarded` near the top) -------------------------------------------------------
extern "C" int ffs_wait(ffs_stream* s, const ffs_frame_result** results, uint32_t* n_results) {
    return guarded(s ? s->ctx : nullptr, [&] { return ffs_wait_impl(s, results, n_results); });
}
extern "C" int ffs_submit_compressed(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes,
                                     uint32_t n_frames, int64_t first_frame_id) {
    return guarded(s ? s->ctx : nullptr, [&] { return ffs_submit_compressed_impl(s, chunks, chunk_bytes, n_frames, first_frame_id); });
}
extern "C" int ffs_ctx_set_mask(ffs_ctx* c, const uint8_t* host_mask) {
    return guarded(c, [&] { return ffs_ctx_set_mask_impl(c, host_mask); });
}
extern "C" int ffs_ctx_get_mask(ffs_ctx* c, uint8_t* host_mask) {
    return guarded(c, [&] { return ffs_ctx_get_mask_impl(c, host_mask); });
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
