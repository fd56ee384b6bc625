// ffs_internal.hpp -- what the translation units of libffs_hip.so share: the structs behind the opaque handles of
// include/ffs_hip.h, error plumbing, and the few functions one unit calls in another.  Host side only; the kernels
// live in kernels_*.hpp, each included by exactly one unit:
//   ffs_context.hip  contexts, masks (kernels_mask.hpp), tuning, streams
//   ffs_submit.hip   launch geometry, the launches of a batch (threshold -> sparse stage), submit entry points,
//                    compressed input (kernels_stream / threshold / extended / ccl / chain / decode)
//   ffs_wait.hip     ffs_wait: overflow re-runs, result assembly, result accessors
//   ffs_stack3d.hip  rotation sweeps: the device-resident 3D stack and the exchange between GPUs (kernels_stack3d)
//   ffs_bench.hip    measurement entry points (kernel timings, memory ceiling, native pipeline loop, sqrt self-test)
// No exception leaves the library (guarded()).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "ffs_hip.h"
#include "ffs_device.h"

using namespace ffsamd;

// ---- errors ---------------------------------------------------------------------------------------------------
extern thread_local std::string g_create_error;   // ffs_last_error(NULL): failures before a context exists

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

// ---- tuning ---------------------------------------------------------------------------------------------------
// Every setting here selects between paths that give the SAME results (A/B partners, fall-backs, capacities that
// tests shrink); they are set per context through ffs_ctx_set_tuning(), never through the environment.  The timing
// experiments that break results exist only in -DFFS_EXPERIMENTS builds (`exp`, read from FFS_EXP_* there).
struct Tuning {
    int threshold_path = 0;     // 0: bright windows -> list -> k_bright_fix; 1: bright windows -> plane -> k_exact (also the
                                //    fall-back when the list overflows); 2: no streaming kernel at all -- EVERY valid pixel is a candidate
                                //    and k_exact gathers its window (the independent partner of `spotfinder --validate`; ~10 ms per frame)
    int ext_first_pass = 2;     // extended algorithm, 16-bit pixels: 2 = streaming kernel, 0 = k_ext_first
    int sparse_stage = 2;       // one launch per batch, a workgroup per frame (k_frame_chain): 3 = always, 2 = unless the stream's previous
                                //    batch held a frame beyond its LDS forest; 1 = four grid-wide kernels
    int sched = 3;              // 3 = the context's streams share one dense, two sparse and one upload HIP stream; 0 = one HIP stream per ffs_stream
    int chain_first = 2;        // while at most n batches are in flight the sparse launch does the bright fix-up and the next
                                //    streaming kernel waits for its start; 0 = never
    int bright_cap = 1 << 20;   // entries of the bright-window list actually used
    int frames_per_group = 1 << 30;   // frames side by side in one super row of the streaming kernels (cap)
    long long target_waves = 16384;   // waves a streaming launch aims for
    int stream_bands = 0;             // > 0: bands of a streaming launch (0: from target_waves, a multiple of eight of at least 72 rows)
    int dense_mask = 0;         // 1: always produce the dense byte mask
    int occupancy_bitmap = 1;   // k_frame_chain reads only the plane segments the occupancy bitmap names
    int direct_records = 1;     // records and counters are written straight into pinned host memory
    int decode_in_dense_stream = 1;   // the decode kernel runs in the dense kernels' stream (0: in the upload stream)
    int ccl_grid = 32;          // workgroups per frame of the grid-wide sparse kernels
    int ext_erode = 2;          // extended algorithm, erosion: 0 = k_ext_erode (a lane per word column, three loads per row), 1 / 2 = k_ext_erode_strips
                                //    (a wave per 62 word columns and 32 / 16 rows, one load per row, neighbours by DPP; non-zero words only
                                //    when the plane was cleared behind the previous batch)
    int ext_e_sparse = 0;       // ... 1 = the signal-region plane is cleared behind the previous batch and the strip erosion stores its non-zero words only
    int ext_fused = 0;          // extended algorithm, 16-bit pixels: 1 = erosion fused into the final pass's tiles (k_ext_erode_final: one launch, the plane
                                //    crosses memory once); 0 = k_ext_erode + k_ext_final.  Measured round 4: the fused kernel is SLOWER (threshold stage 0.65
                                //    against 0.55 ms per 32 frames, profiles/r04d_ext_fused_ab.txt): every tile starts with a chain of dependent plane loads
                                //    that its gathers then wait behind, 17 000 times per batch -- kept as an A/B partner, parity-tested
    int ext_rest_aside = 0;     // extended algorithm: 1 = erosion + final pass in the batch's sparse stream, beside the next batch's first pass; 0 = in the
                                //    dense stream (measured round 4: no gain -- a CU full of first-pass waves has neither LDS nor registers left for the
                                //    final pass's workgroups, so the kernels take turns either way: profiles/r04b_ext_streams_ab.txt)
    int band_taper = 0;         // streaming kernels: the last two bands per XCD are this many per cent as tall as the others (0 = uniform bands)
    int rows_ahead = 3;         // rows of loads a wave of the streaming kernels keeps in flight (16-bit pixels: 2, 3 or 4; 32-bit: 2 or 3).  Round 5: three.
                                //    Alone the kernel is the same with two (round 4's measurement, and why it was two); in the pipeline, beside the
                                //    band launches, three is 5 % faster (0.292-0.296 against 0.309-0.312 ms: profiles/r05zj_rows_ahead_ab.log)
    int device_lists = 2;       // the strong-pixel lists stay on the device after a batch: 1 = always, 0 = only when the host asked for them
                                //    (want_strong_list), 2 = also while a 3D stack of the process is alive (ffs_stack3d_add_batch reads them)
    int strong_log = 1;         // 16-bit standard path: the streaming kernel appends its strong groups to per-wave logs and the one-launch sparse
                                //    stage merges them (kernels_chain.hpp, LOG) instead of scattering plane bytes, counters and occupancy bits;
                                //    0 = the bit plane (also what dense frames, tall frames and the other algorithms and paths take)
    int chain_runs = 1;         // sparse_stage 2: frames beyond the LDS forest of pixels stay in the one launch when their RUNS fit
                                //    (16-bit pixels, rows up to 16383 pixels); 0 = such batches take the four grid-wide kernels; 2 = runs for every frame
    int sparse_bands = 1;       // standard path with wave logs, lists not asked for: the sparse stage in small workgroups (kernels_band.hpp: a wave per
                                //    band of a frame + a merge per frame) instead of k_frame_chain's one workgroup per frame: 1 = with one or with four
                                //    and more batches in flight (measured: ffs_submit.hip), 2 = always, 0 = never
    int dense_overlap = 0;      // wave-log path: 1 = consecutive streaming kernels on two HIP streams, handed over by a value the launch's last workgroup
                                //    writes as it starts (hipStreamWaitValue32); 0 = one dense stream, a barrier between its dispatches.  Measured round 5:
                                //    the idea works in isolation (tools/ubench/wait_value.hip: 159 -> 146 us per launch of 15 064 sleeping waves) and LOSES
                                //    8-10 % in the pipeline (0.335-0.35 against 0.307-0.313 ms per step, four or eight hardware queues:
                                //    profiles/r05t_dense_overlap_ab.log) -- the next kernel's first waves and the band launches then fight over the same
                                //    freed slots, and each streaming kernel takes 0.325 instead of 0.298 ms.  Off; kept as the A/B partner
    int stream_prio = 1;        // 1: the 16-bit streaming kernel's waves run at issue priority 3 (s_setprio), ahead of the band waves that share their SIMDs
                                //    (+1 % on the driver-style line, six alternating pairs: profiles/r05zm_stream_prio_ab.log); 0: default priority
    int assembly_threads = 7;   // helper threads that assemble a batch's result arrays beside the caller (made with the context's first large batch)
    int wait_ahead = 1;         // a thread of the context assembles each batch's result arrays as soon as the GPU has finished it (0: ffs_wait does)
    int sparse_priority = 0;    // priority of the context's sparse HIP streams: 0 = highest, 1 = lowest, 2 = the dense stream's
#ifdef FFS_EXPERIMENTS
    struct Exp {
        int k1_debug = 0, chain_skip = 0, chain_stop = 0, dummy_us = 0, dummy_wg = 32, dummy_threads = 1024, dummy_lds = 0;
    } exp;
#endif
};

struct ffs_stack3d;

// A few helper threads per context for ffs_wait's result assembly (wire records -> boxes, reflections, centre rows): one thread
// moves a batch's 7 MB (45 000 components of 32 Eiger frames) in 0.21 ms -- two thirds of the time the GPU takes for the batch, and
// the last waits of a run pay it on the clock.  Frames are independent (their slices of the output arrays follow from the
// per-frame summaries), so the caller and the helpers take frames off a shared counter.  Created on the first large batch; a
// wait that finds the pool busy (another stream's wait is using it) assembles on its own.
struct AssemblyPool {
    std::vector<std::thread> threads;
    std::mutex mu;                       // guards job / generation / stop
    std::condition_variable cv;
    std::mutex owner;                    // one wait at a time uses the helpers
    const std::function<void(uint32_t)>* job = nullptr;
    uint32_t n_items = 0;
    std::atomic<uint32_t> next{0}, done{0};
    std::atomic<int> active{0};          // helpers inside the current job (run() does not return before they have left it)
    uint64_t generation = 0;
    bool stop = false;
    void start(int n_threads);
    void run(uint32_t n, const std::function<void(uint32_t)>& fn);   // fn(i) for i in [0, n), on the caller and the helpers
    void stop_and_join();                // the helpers leave (after the job they are in) and are joined; run() then works on the caller alone
    ~AssemblyPool();
};

// A staging area the DMA engines can read: anonymous memory on transparent huge pages, registered with the runtime
// (tools/ubench/pin_cost.hip: 17 ms per 256 MB against 54 + 26 ms to allocate and free the same with hipHostMalloc, same
// 57 GB/s to the device); hipHostMalloc when registering is refused.
struct PinnedBuf {
    uint8_t* p = nullptr;      // the usable area (2 MiB aligned when mapped)
    size_t bytes = 0;
    void* map_base = nullptr;  // mmap'ed region (null: p came from hipHostMalloc)
    size_t map_len = 0;
};
PinnedBuf pinned_alloc(size_t bytes);   // p == nullptr on failure
void pinned_free(PinnedBuf& b);

// One thread per context that assembles the results of the context's batches as the GPU finishes them (ffs_wait.hip): ffs_wait then
// finds a batch's arrays ready instead of spending 0.1-0.2 ms on them -- which the waits at the END of a run, one behind the other
// with nothing left to hide them, paid on the clock.  Only batches that need nothing else from the host (no overflow, no list or
// mask copies); everything else is left to the caller's ffs_wait as before.
struct AheadThread {
    std::thread th;
    std::mutex mu;                 // guards q, stop and every stream's ahead_state
    std::condition_variable cv_work, cv_done;
    std::deque<ffs_stream*> q;     // registered batches, in submit order
    bool stop = false;
};

struct ffs_ctx {
    int device = 0;
    Tuning tune;
    Layout L{};
    int pixel_bytes = 2;
    uint32_t max_batch = 1;
    uint32_t cap = 0;       // strong pixels per frame
    uint32_t max_comp = 0;  // components per frame
    int n_tiles = 0;
    ffs_params params{};
    uint8_t* d_maskbits = nullptr;
    uint8_t* d_ginfo = nullptr;  // per-group mask bits + window-count bounds (kernels_stream.hpp)
    uint8_t* d_mmap = nullptr;   // per-pixel window counts
    hipStream_t dense_st = nullptr;  // sched 3: the one stream of the dense kernels
    // ... and its partner (tuning "dense_overlap", off: measured slower in the pipeline): the streaming kernels of the wave-log path take
    // the two alternately, each behind a wait for the value the previous launch's last workgroup writes into `d_handoff` (signal
    // memory) as it starts.  Both are made by the first launch that asks.  dense_mu: one launch at a time decides.
    hipStream_t dense_st2 = nullptr;
    uint32_t* d_handoff = nullptr;
    std::mutex dense_mu;
    uint32_t handoff_seq = 0;
    int handoff_last = -1;           // which of the two streams took the last launch of the chain
    hipStream_t up_st = nullptr;     // ... and the one stream of uploads and decoding
    hipStream_t sparse_st[2] = {nullptr, nullptr};  // ... the sparse launches of the context's streams, alternating
    int n_streams_made = 0;
    std::mutex stream_mu;            // guards the lazy creation of the shared streams, the stack pool, the event ring and the two lists below
    // What the context's users have created on it and not destroyed yet: ffs_ctx_destroy closes these first (a stream or a stack
    // keeps a pointer to its context), and the process's exit handler finds in-flight work through them (lifecycle, below).
    std::vector<ffs_stream*> live_streams;
    std::vector<ffs_stack3d*> live_stacks;
    std::vector<ffs_stack3d*> stack_pool;   // destroyed 3D stacks kept with their buffers for the next sweep (stream_mu)
    // Pinned host memory costs ~170 ms per GB to allocate and ~100 ms per GB to free, and the runtime serialises both
    // across threads (tools/ubench/alloc_cost.hip): the staging buffers of destroyed streams are kept for the next
    // stream of the context and freed with it (stream_mu).
    std::vector<struct PinnedBuf> pinned_pool;
    std::atomic<int> inflight{0};    // batches between submit and wait, over all ffs_streams of the context
    // Start events of the sparse launches (they ride on the dispatch): the next streaming kernel lets the newest one get
    // its CUs first.  The events belong to the CONTEXT (created with the first stream, destroyed with the context), so a
    // thread may wait on one while another thread destroys the ffs_stream that recorded it.
    static constexpr int kChainEvents = 16;
    hipEvent_t chain_ev[kChainEvents] = {};
    std::atomic<uint32_t> chain_ev_next{0};     // slots handed out so far
    std::atomic<int> chain_ev_newest{-1};       // slot of the newest recorded start, -1: none yet
    bool chain_ok = false;           // k_frame_chain may use its dynamic LDS on this device
    std::atomic<AssemblyPool*> assembly{nullptr};   // helper threads of ffs_wait (created on first use under stream_mu; read without it)
    std::atomic<AheadThread*> ahead{nullptr};       // assembles results as batches complete (created on first use under stream_mu)
    ThreadError err;  // the calling thread's most recent error on any context
};

struct OverflowFrame;

constexpr uint32_t kBrightCap = 1u << 20;  // entries of the bright-window list per batch (8 MB)

struct ffs_stream {
    ffs_ctx* ctx = nullptr;
    uint32_t max_batch = 1;   // frames per submit
    uint32_t cap = 0;         // strong pixels per frame the lists hold
    uint32_t max_comp = 0;    // components per frame the record buffers hold
    ffs_stream* big = nullptr;               // one-frame stream with room for frames that exceed cap / max_comp
    std::vector<OverflowFrame> ovf;          // such frames of the last batch, re-run on `big`
    int force_path = -1;                     // >= 0: threshold path of the next enqueue (bright-list overflow -> 1)
    bool lists_valid = true;                 // the last batch left its strong-pixel lists on the device
    bool plane_once = false;                 // the next enqueue takes the plane (a batch the logs could not serve is run again)
    bool log_off = false;                    // the wave logs could not serve a batch of this stream (dense frames, a log overflow): the plane from then on
    uint2* d_wlog = nullptr;                 // wave logs of the streaming kernel (allocated on first use, sized for the launch geometry)
    uint32_t* d_wlog_n = nullptr;
    uint4* d_wpix = nullptr;
    size_t wlog_waves = 0;
    // the sparse stage in small workgroups (kernels_band.hpp): what the band waves hand to the per-frame merge (allocated on first use)
    uint4* d_band_hdr = nullptr;
    uint8_t* d_band_acc = nullptr;
    uint32_t* d_band_seam = nullptr;
    uint32_t band_slots = 0;                 // (frame, band) pairs the three buffers hold
    uint32_t band_backoff = 0;               // batches this stream still sends through k_frame_chain after a band overflowed its plan (flag 128: dense data)
    bool bands_once_off = false;             // the next enqueue takes k_frame_chain (the batch that raised flag 128 is run again)
    bool band_mode = false;                  // this batch's sparse stage is k_band_cc + k_frame_merge
    uint32_t path_bits = 0;                  // which launches the last batch took (ffs_stream_last_path)
    uint32_t reruns = 0;                     // times ffs_wait ran the last batch again (a plan that did not hold it)
    bool force_grid = false;                 // the next enqueue takes the grid-wide sparse kernels (a frame's runs overflowed the one launch)
    bool runs_overflowed = false;            // ... and dense batches of this stream keep taking them
    uint32_t *d_pack_k = nullptr, *d_pack_i = nullptr;  // a batch's lists packed end to end for another device's 3D stack
    StackSlice *d_pack_tab = nullptr, *h_pack_tab = nullptr;
    hipEvent_t ev_pack = nullptr;            // the packed lists are ready on the source device
    hipEvent_t ev_sent = nullptr;            // ... and have been copied out of the pack buffers (an event of device ev_sent_dev, the stack's)
    int ev_sent_dev = -1;
    bool sent_pending = false;
    hipStream_t st = nullptr;    // threshold kernels (+ H2D)
    hipStream_t st_up = nullptr; // uploads + decode; == st unless the dense kernels of the context share one stream
    bool st2_shared = false;
    bool st_shared = false;      // st is the context's dense stream (not ours to destroy)
    hipStream_t st2 = nullptr;   // compaction + connected components + D2H; == st unless the context has sparse streams
    hipEvent_t ev[7] = {};   // [6]: the compressed chunks and their block table are on the device
    // device (one allocation, d_slab, carved up at creation; the buffers of rarely used paths are allocated on first use)
    uint8_t* d_slab = nullptr;
    uint8_t* d_img = nullptr;
    uint8_t* d_bits = nullptr;
    uint8_t* d_sbytes = nullptr;
    uint8_t *d_dplane = nullptr, *d_eplane = nullptr;  // extended algorithm only (allocated on first use)
    // The first-pass plane must be all zero when the streaming kernel starts (it writes non-zero bytes only).  Two planes take
    // turns: the one the previous batch used is cleared in the sparse stream behind this batch's sparse launch -- done before
    // this batch's last event, i.e. before the stream's next submit -- instead of by a fill in the dense stream ahead of every
    // first pass; the last batch's plane stays readable (ffs_stream_debug_bitplane, --writeout).
    // Round 4: the signal-region plane has a twin too (the erosion then stores only the words that hold a pixel of the region); a
    // first-pass plane and a signal-region plane are ONE allocation (d_ext_pair), so one fill clears both.
    uint8_t* d_dplane2 = nullptr;
    uint8_t* d_eplane2 = nullptr;
    uint8_t* d_ext_pair[2] = {nullptr, nullptr};
    bool dplane2_clean = false;                        // the first-pass plane the NEXT batch takes is zero
    bool eplane2_clean = false;                        // ... and so is the signal-region plane behind it
    bool ext_e_clean = false;                          // this batch's signal-region plane is zero (make_threshold_args passes it on)
    uint8_t* d_comp = nullptr;                         // compressed chunks (allocated on first use)
    uint2 *d_tab = nullptr, *h_tab = nullptr;          // per-block (offset, length) tables
    uint32_t dec_blocks = 0, dec_last = 0, dec_tail = 0, dec_block_elems = 0;
    // Without direct records: copied back speculatively with the counts (one wait instead of two): room for the most
    // records per frame seen so far on this stream, +25 %; ffs_wait fetches the rest if a batch exceeds it.
    uint32_t spec_recs_per_frame = 256;
    uint64_t spec_recs_copied = 0;
    std::thread job;          // ffs_submit_compressed's helper (block index + launches); joined by ffs_wait
    int job_rc = 0;
    std::string job_err;
    uint32_t *d_tile_counts = nullptr, *d_num_strong = nullptr, *d_row_off = nullptr;
    uint2* d_bright = nullptr;  // pixels the streaming kernels hand to k_bright_fix; their count sits behind the tile counts
    uint32_t *d_list_k = nullptr, *d_list_i = nullptr, *d_parent = nullptr;
    uint32_t *d_n_comp = nullptr, *d_overflow = nullptr, *d_summary = nullptr;
    CompAcc2* d_acc2 = nullptr;          // accumulators at the root's list index
    uint32_t* d_chunk_roots = nullptr;
    ReflOut* d_recs = nullptr;
    // pinned host
    uint8_t* h_img = nullptr;      // pinned staging: allocated on first use (ensure_host_staging), sized by what is asked for
    size_t h_img_bytes = 0;
    PinnedBuf h_img_buf;           // ... and how it was obtained
    size_t d_comp_bytes = 0;
    uint32_t* h_counts = nullptr;  // [max_batch] num_strong | [max_batch] n_comp | [max_batch*8] summary | [1] overflow | [max_batch] per-frame flags
    ReflOut* h_recs = nullptr;
    uint32_t* d_occ = nullptr;     // [max_batch][occ_frame_words] occupancy of the strong plane (one bit per 16-byte segment)
    uint32_t* h_counts_dev = nullptr;  // device-side address of h_counts (k_frame_chain writes the counters itself)
    bool ev1_pending = false;      // ev[1] (start of the threshold stage) has not been recorded yet for this batch
    bool ev3_is_ev4 = false;       // one event behind the sparse launch (k_frame_chain leaves nothing to copy)
    bool dev_input = false;        // this batch's frames were on the device already (ffs_submit_device): no upload, no ev[0]
    bool dense_valid = false;      // the byte masks of the last batch were produced
    bool chain_mode = false;       // this batch went through k_frame_chain: records at frame * max_comp, flags per frame
    ReflOut* h_recs_dev = nullptr;  // device-side address of h_recs when the records are written straight to the host
    bool direct_recs = false;
    bool bits_cleared = false;  // the last batch's compaction zeroed the strong plane again (the streaming kernels' invariant)
    bool bits_dirty = false;    // the strong plane may hold bits: the streaming kernels need it zeroed first
    bool counts_dirty = true;   // the per-tile counts (+ bright-list count) may be non-zero: the streaming kernels add into them
    bool occ_dirty = false;     // the occupancy bitmap may hold bits nobody will consume
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
    bool timings_stale = false;              // the stage times of the last batch are still in its events (ffs_stream_timings reads them out)
    hipEvent_t timing_last = nullptr;
    // results
    std::vector<ffs_frame_result> results;
    std::vector<ffs_box> boxes;
    std::vector<ffs_reflection> refls;
#ifdef FFS_EXPERIMENTS
    unsigned long long *h_phase_ts = nullptr, *h_phase_ts_dev = nullptr;   // device timestamps of the sparse launch's phases (pinned, [B][8])
    double phase_sum[8] = {};       // accumulated phase durations, us (printed when the stream is destroyed; FFS_EXP_CHAIN_TS)
    unsigned long long phase_n = 0;
#endif
    std::vector<float> centres;   // (frame id bits, com_x, com_y, com_z) per reflection of the last batch, filled with `refls` (ffs_stream_spot_centres)
    // the same four for the batch in flight, filled by the context's AheadThread and swapped in by ffs_wait (what the last ffs_wait
    // returned stays valid until the next one)
    std::vector<ffs_frame_result> results_n;
    std::vector<ffs_box> boxes_n;
    std::vector<ffs_reflection> refls_n;
    std::vector<float> centres_n;
    int ahead_state = 0;          // 0: not registered, 1: queued / being assembled, 2: assembled into the *_n arrays, 3: left to the caller (AheadThread::mu)
};

// Results of a frame that did not fit the stream's lists, from its re-run on the one-frame stream
struct OverflowFrame {
    uint32_t frame = 0;
    ffs_frame_result res{};
    std::vector<ffs_box> boxes;
    std::vector<ffs_reflection> refls;
    std::vector<uint32_t> k, inten;
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

// ---- small helpers ------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline uint32_t occ_frame_words(const Layout& L) { return (uint32_t)(((uint64_t)L.H * (L.mpitch / 16) + 31) / 32 + 2); }  // (+2: the chain reads a word ahead)
// per-tile counts | ... | [last - 1] workgroups of k_frame_chain through with the bright list | [last] entries of the bright list;
// a multiple of 256 bytes so that one fill clears it
static inline size_t tile_counts_bytes(const ffs_stream* s) { return (((size_t)s->max_batch * s->ctx->n_tiles + 2) * 4 + 255) / 256 * 256; }
static inline void mark_busy(ffs_stream* s) {
    if (!s->busy) ++s->ctx->inflight;
    s->busy = true;
}
static inline void mark_idle(ffs_stream* s) {
    if (s->busy) --s->ctx->inflight;
    s->busy = false;
}
template <typename T>
static hipError_t dmalloc(T** p, size_t n_bytes) {
    return hipMalloc(reinterpret_cast<void**>(p), n_bytes + 256);
}

// ---- lifecycle: which handles are alive ---------------------------------------------------------------------
// Every handle the ABI has given out is in a process-wide registry until it is destroyed -- by its own destroy call, or by the
// ffs_ctx_destroy of its context, which closes a context's streams and stacks before the context itself.  A destroy call on a handle
// that is not (or no longer) in the registry does nothing, so the ORDER in which a caller (a binding's finalisers, a C++ driver's
// unwinding, a test that leaks) lets go of contexts, streams and stacks cannot reach freed memory.  When the process exits with
// handles alive, the library's own exit handler -- registered after the HIP runtime has initialised, so it runs BEFORE the runtime's
// handlers -- joins the helper threads and waits for the work in flight (kernels that write into pinned host memory); every destroy
// call after that is a no-op and the runtime's own teardown takes the memory.  DESIGN.md section 10c.
enum HandleKind { kHandleCtx = 0, kHandleStream = 1, kHandleStack = 2 };
void handle_add(HandleKind kind, const void* h);
bool handle_take(HandleKind kind, const void* h);   // removes h; false: not a live handle (destroyed already, or the process is exiting)
bool handle_live(HandleKind kind, const void* h);
bool process_exiting();
// A stream handle that its context's ffs_ctx_destroy (or an earlier ffs_stream_destroy) has closed is refused by submit and wait
// with FFS_ERR_INVALID and a text in ffs_last_error(NULL), instead of being followed into freed memory.
static inline bool stream_handle_ok(const ffs_stream* s) {
    if (handle_live(kHandleStream, s)) return true;
    g_create_error = "stale ffs_stream handle: the stream was destroyed (with its context, or by ffs_stream_destroy)";
    return false;
}
void stream_destroy_internal(ffs_stream* s);         // what ffs_stream_destroy does once the handle is out of the registry

// ---- functions one unit calls in another ----------------------------------------------------------------------
// ffs_context.hip
int stream_create_sized(ffs_ctx* c, uint32_t max_batch, uint32_t cap, uint32_t max_comp, ffs_stream** out);
int ensure_host_staging(ffs_stream* s, size_t bytes);   // pinned staging of at least `bytes` (contents are not kept when it grows)
size_t default_staging_bytes(const ffs_stream* s);      // max_batch raw frames (+ the slack incompressible chunks need)
// ffs_submit.hip
bool chain_prepare_device();   // asks for k_frame_chain's dynamic LDS on the current device; false: use the four kernels
ThresholdArgs make_threshold_args(ffs_stream* s, const void* img, size_t pitch, size_t fstride, uint32_t n_frames);
int check_layout(ffs_stream* s, size_t pitch, size_t fstride, uint32_t n_frames);
int ensure_extended_buffers(ffs_stream* s);
int enqueue_batch(ffs_stream* s, const void* d_img, size_t pitch, size_t fstride, uint32_t n, const ffs_params* snapshot = nullptr);
// one launch of the threshold stage's dense kernel on s->st with HIP events on the dispatch itself (either may be null),
// and of the kernel that follows it (k_bright_fix / k_exact; extended: erosion + final pass) -- what ffs_bench_threshold times
extern std::atomic<int> g_live_stacks;   // 3D stacks alive in the process (ffs_stack3d.hip)
bool wave_logs_for(ffs_stream* s, ThresholdArgs& a, uint32_t n_frames);
void bench_launch_dense(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames, hipEvent_t start, hipEvent_t stop);
void bench_launch_rest(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames);
// ffs_wait.hip
int ffs_wait_impl(ffs_stream* s, const ffs_frame_result** results, uint32_t* n_results);
void ahead_register(ffs_stream* s);   // the batch just enqueued may be assembled ahead of its ffs_wait
int ahead_take(ffs_stream* s);        // waits for the AheadThread to be done with the stream's batch; its verdict (0, 2 or 3), state reset
void ahead_stop(ffs_ctx* c, bool destroy);   // the thread leaves and is joined (context destroy, process exit)
// ffs_stack3d.hip
void stack3d_free(ffs_stack3d* st);
void gather_scratch_free(ffs_stream* s);   // what ffs_multi_gather_rows allocated for the stream, if anything
