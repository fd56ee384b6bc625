/*
 * ffs_hip.h -- C ABI of libffs_hip.so: the MI355X (gfx950) spot-finder hot path.
 *
 * This is the drop-in boundary for the reference's per-frame device work and the
 * host connected-components stage that follows it.  Each entry point names the
 * reference interface it replaces (paths relative to the reference checkout).
 * Plain C: opaque handles, plain pointers and sizes, int status returns
 * (0 = FFS_OK, negative = error; ffs_last_error() gives the text).  No
 * exceptions cross this boundary and no torch / HIP types appear in it.
 *
 * Data flow of one batch:
 *   frames (u16/u32: dense host rows, pitched device rows, or bitshuffle-LZ4 chunks decoded on the GPU)
 *     -> ONE streaming threshold kernel per batch: exact integer 7x7 window sums, a conservative group
 *        screen, and the oracle's float64 predicate (baseline/spotfinder/standalone.cc:113-174) decided
 *        in its drain -- replaces kernels/thresholding.cu:145-234; the strong mask stays a bit plane
 *     -> ONE sparse launch per batch, a workgroup per frame: strong-pixel compaction, union-find connected
 *        components, centroids and filters (replaces connected_components.cc:17-139, 207-266)
 *     -> per-frame counters and 40-byte records written straight into pinned host memory
 */
#ifndef FFS_HIP_H
#define FFS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libffs_hip.so is built with -fvisibility=hidden: what this header declares is all it exports */
#pragma GCC visibility push(default)

#define FFS_OK 0
#define FFS_ERR_INVALID (-1)   /* bad argument / state */
#define FFS_ERR_DEVICE (-2)    /* HIP runtime error */
#define FFS_ERR_NOMEM (-3)     /* host or device allocation failed */
#define FFS_ERR_OVERFLOW (-4)  /* a frame produced more strong pixels / spots than the context was sized for */
#define FFS_ERR_NODEVICE (-5)  /* no usable GPU */

typedef struct ffs_ctx ffs_ctx;
typedef struct ffs_stream ffs_stream;
typedef struct ffs_stack3d ffs_stack3d;

/* Algorithm parameters.  Defaults (ffs_default_params) are the oracle's,
 * baseline/spotfinder/standalone.cc:16-20: 7x7 window, min_count 2, nsig_b 6,
 * nsig_s 3, threshold 0; the GPU reference's launch wrapper defaults
 * (spotfinder/spotfinder.cuh:18-20) differ only in min_count = 3. */
typedef struct {
    int32_t min_count;        /* window needs >= this many valid pixels */
    double nsig_b;            /* background (dispersion) significance */
    double nsig_s;            /* signal significance */
    double threshold;         /* centre pixel must be > threshold */
    int64_t max_valid;        /* centre pixel must be <= max_valid (thresholding.cu:208-215);
                                 < 0 = no test (oracle behaviour) */
    uint32_t min_spot_size;   /* --min-spot-size, spotfinder.cc:318-322 (default 3) */
    uint32_t min_spot_size_3d;/* --min-spot-size-3d, :324-328 (default 3) */
    float max_peak_centroid_separation; /* :330-336 (default 2.0) */
    int32_t want_reflections; /* compute find_2d_components output per frame (spotfinder.cc:919-933) */
    int32_t want_strong_list; /* return the sparse strong-pixel list (k, intensity) per frame */
    int32_t want_strong_mask; /* produce and return the dense W*H byte mask per frame (reference D2H, :887-897); 0 (default): the
                               * strong mask stays a bit plane on the device and only lists / boxes / reflections come back */
    int32_t algorithm;        /* FFS_ALGO_DISPERSION (default) or FFS_ALGO_DISPERSION_EXTENDED:
                                 `--algorithm`, spotfinder.cc:338-342, 572-590 */
    int32_t extended_flavour; /* extended only.  0 = baseline.cpp:730-761 rules (default); 1 = where the
                                 device kernels differ structurally: erosion skips masked neighbours
                                 (erosion.cu:101-105), second pass needs n > 0 (thresholding.cu:472) */
} ffs_params;

#define FFS_ALGO_DISPERSION 0
#define FFS_ALGO_DISPERSION_EXTENDED 1

void ffs_default_params(ffs_params *p);

/* struct Reflection, spotfinder/connected_components/connected_components.hpp:27-30 */
typedef struct {
    uint32_t l, t, r, b;
    int32_t num_pixels;
} ffs_box;

#define FFS_REFL_FILTERED_SIZE 1u /* removed by min_spot_size (connected_components.cc:213-222) */
#define FFS_REFL_FILTERED_SEP 2u  /* removed by peak-centroid distance (:224-233) */

/* class Reflection3D, connected_components.hpp:32-260 (what its getters and
 * center_of_mass() / peak_centroid_distance() return). */
typedef struct {
    uint32_t x_min, x_max, y_min, y_max;
    int32_t z_min, z_max;
    int32_t num_pixels;
    float com_x, com_y, com_z;
    uint32_t peak_x, peak_y;
    int32_t peak_z;
    uint32_t peak_intensity;
    float peak_centroid_distance;
    uint32_t flags;
    uint64_t sum_intensity;
} ffs_reflection;

/* What the reference's worker has in hand after ConnectedComponents(...)
 * (spotfinder.cc:901-933) for one frame.  Pointers stay valid until the next
 * ffs_wait() on the same stream. */
typedef struct {
    int64_t frame_id;
    uint32_t num_strong_pixels;          /* get_num_strong_pixels() */
    uint32_t num_strong_pixels_filtered; /* get_num_strong_pixels_filtered() */
    uint32_t n_components;               /* "Extracted {} spots", connected_components.cc:119 */
    uint32_t n_boxes;                    /* boxes.size() after the min-size filter = JSON n_spots_total */
    const ffs_box *boxes;                /* label order */
    uint32_t n_reflections;              /* find_2d_components(): after both filters, label order */
    const ffs_reflection *reflections;   /* NULL unless want_reflections */
    uint32_t n_filtered_size, n_filtered_sep;
    const uint32_t *strong_k;            /* ascending linear index y*W+x; NULL unless want_strong_list */
    const uint32_t *strong_intensity;
    const uint8_t *strong_mask;          /* dense W*H, NULL unless want_strong_mask */
} ffs_frame_result;

/* ---- devices (replaces CUDAArgumentParser --list-devices / -d, src/ffs/cuda_arg_parser.cc:30-61) */
int ffs_device_count(void);
int ffs_device_name(int device, char *buf, size_t buflen);
int ffs_device_total_mem(int device, uint64_t *bytes);

/* ---- context: one per (GPU, detector geometry) --------------------------------------------- */
/* pixel_bytes: 2 (uint16) or 4 (uint32) -- replaces the compile-time pixel_t switch
 * (h5read/include/h5read.h:16-20, spotfinder/CMakeLists.txt:45-73).
 * max_batch: frames per submit.  max_strong_per_frame: strong pixels per frame the batch lists are sized for,
 * 0 = default min(W*H, 2^18) (and min(that, 65536) components).  Not a limit on the data: a frame that exceeds
 * either is run again on its own with room for it inside ffs_wait() (the reference's std::map of signals,
 * connected_components.cc:24-32, has no bound), at the price of that extra pass. */
int ffs_ctx_create(int device, uint32_t width, uint32_t height, int pixel_bytes,
                   uint32_t max_batch, uint32_t max_strong_per_frame, ffs_ctx **out);
/* Lifetime rules (the reference: RAII members of one worker thread, spotfinder.cc:729-742; CUDA_CHECK aborts by exception,
 * include/cuda_common.hpp:28-45 -- here nothing aborts):
 *  - ffs_ctx_destroy() first closes the streams and 3D stacks that were created on the context and are still alive (waiting for
 *    their work in flight), then the context.  Their handles are dead from then on.
 *  - every destroy call is idempotent: on a handle that was destroyed already (by its own destroy call or with its context) it does
 *    nothing, so a caller's teardown -- finalisers of a binding, unwinding of a driver -- may let go of contexts, streams and stacks
 *    in ANY order.  ffs_submit* / ffs_wait on such a handle return FFS_ERR_INVALID.  (Handles are compared by address: do not keep a
 *    dead handle across the creation of new ones.)
 *  - no other thread may be inside a call on a handle, or on a child of a context, while it is being destroyed.
 *  - a process may exit (exit(), return from main, an interpreter shutting down) with handles alive and batches in flight: the
 *    library's exit handler, which runs before the HIP runtime's own, joins its helper threads and drains the devices; destroy calls
 *    that arrive after that (static destructors, late finalisers) do nothing.  _exit() / quick_exit() skip it, like any handler. */
void ffs_ctx_destroy(ffs_ctx *ctx);
/* Text of the calling thread's most recent error (kept per thread: worker threads drive their own
 * streams of one context).  ctx may be NULL: last error of ffs_ctx_create. */
const char *ffs_last_error(const ffs_ctx *ctx);

/* upload_mask(), spotfinder/spotfinder.cc:61-108: host_mask = W*H bytes, nonzero = valid,
 * or NULL for "all valid" (the reference's cudaMemset(1) branch). */
int ffs_ctx_set_mask(ffs_ctx *ctx, const uint8_t *host_mask);

/* call_apply_resolution_mask(), spotfinder/kernels/masking.cuh:82-97 + masking.cu:37-147:
 * clears mask bits whose d-spacing is outside [dmin, dmax] (<= 0 = unbounded). */
int ffs_ctx_apply_resolution_mask(ffs_ctx *ctx, float wavelength, float distance_m,
                                  float beam_center_x_px, float beam_center_y_px,
                                  float pixel_size_x_m, float pixel_size_y_m, float dmin,
                                  float dmax);
/* Read the current mask back (W*H bytes, 0/1) -- what --writeout copies for mask_calculated.png */
int ffs_ctx_get_mask(ffs_ctx *ctx, uint8_t *host_mask);

int ffs_ctx_set_params(ffs_ctx *ctx, const ffs_params *p);

/* Selects between paths that give the SAME results (A/B partners, fall-backs, capacities that tests shrink) --
 * per context, never through the environment; nothing here can change a result.  Keys (default):
 *   "threshold_path"   (0) 0 = windows the streaming kernel cannot vouch for go onto a list (fix-up kernel),
 *                          1 = they are marked in the plane and an exact kernel filters it (also the fall-back
 *                          when that list overflows), 2 = no streaming kernel at all: the window of EVERY valid pixel is
 *                          gathered from memory and decided by the exact kernel (~10 ms per Eiger frame: the independent
 *                          partner `spotfinder --validate` compares the hot path with, spotfinder.cc:1012-1053)
 *   "ext_first_pass"   (2) extended algorithm, 16-bit pixels: 2 = streaming kernel, 0 = plain one-pixel-per-lane kernel
 *   "sparse_stage"     (2) one launch per batch, a workgroup per frame: 3 = always, 2 = unless the stream's previous batch
 *                          held a frame with more strong pixels than that workgroup's LDS holds; 1 = four grid-wide kernels
 *   "device_lists"     (2) a batch leaves its strong-pixel lists on the device: 1 = always, 0 = only when the host asked for them
 *                          (want_strong_list), 2 = also while a 3D stack of the process is alive (ffs_stack3d_add_batch reads them:
 *                          create the stack before submitting).  Without them the sparse launch saves two scattered stores per pixel
 *   "strong_log"       (1) 16-bit standard path: the streaming kernel appends its strong groups to per-wave logs which the
 *                          one-launch sparse stage merges (dense stores; the kernel's time no longer depends on where the
 *                          stream's buffers lie); 0 = plane bytes, counters and an occupancy bitmap as in rounds 1-3b
 *   "chain_runs"       (1) with "sparse_stage" 2: such dense batches of 16-bit frames stay in the one launch, its union-find
 *                          over runs of strong pixels instead of pixels (up to 16384 runs per frame); 0 = they take the grid-wide kernels;
 *                          2 = every frame of 16-bit pixels goes over runs (A/B partner: no faster on sparse frames)
 *   "sparse_bands"     (1) standard path with wave logs when nobody reads the pixel lists or the byte mask: the sparse stage in small
 *                          workgroups -- a wave per band of a frame, then a merge per frame (round 5) -- instead of the one workgroup per
 *                          frame, which holds a whole CU: 1 = with one or with four and more batches in flight, and always on a context with
 *                          four or more streams (where it measures faster), 2 = always, 0 = never.  A band beyond its plan (768 strong pixels, 256 components) sends the batch back
 *                          through the one-workgroup launch inside ffs_wait
 *   "wait_ahead"       (1) a thread of the context turns each batch's records into the result arrays as soon as the GPU has finished
 *                          it, so ffs_wait finds them ready (0: ffs_wait does it, as in rounds 1-4)
 *   "stream_prio"      (1) bit 1: the 16-bit streaming kernel's waves run at issue priority 3 (s_setprio), ahead of the band waves on their
 *                      SIMDs; bit 2: the band and merge waves do (measured slower: DESIGN.md section 3.4c); 0..3
 *   "dense_overlap"    (0) 1 = consecutive streaming kernels of the wave-log path on two HIP streams, handed over by a value the launch's
 *                          last workgroup writes as it starts (hipStreamWaitValue32) instead of the queue's barrier between two dispatches;
 *                          works in isolation, measured 8-10 % slower in the pipeline: off, kept as the A/B partner
 *   "sparse_priority"  (0) priority of the context's two sparse HIP streams: 0 = highest, 1 = lowest, 2 = the dense stream's (before
 *                          the first stream is created; measured: 0 and 1 alike, 2 costs 15 %)
 *   "sched"            (3) 3 = shared dense / sparse / upload HIP streams per context, 0 = one per ffs_stream
 *                          (before the first stream is created)
 *   "direct_records"   (1) records written straight into pinned host memory (before the first stream is created)
 *   "chain_first" (2), "bright_cap" (2^20), "frames_per_group", "target_waves" (16384), "stream_bands" (0: from target_waves; > 0: that many bands, any number), "dense_mask" (0),
 *   "occupancy_bitmap" (1), "decode_in_dense_stream" (1), "rows_ahead" (3: rows of loads a streaming wave keeps in flight), "ccl_grid" (32),
 *   "assembly_threads" (7: helper threads that build a batch's result arrays; 3, 12 and 15 measure the same): see DESIGN.md.
 *   "band_taper" (0), "ext_rest_aside" (0), "ext_fused" (0): round 4's A/B partners (tapered bands of the streaming kernels;
 *                          extended algorithm: erosion + final pass in the sparse stream / fused into one kernel) -- measured, no
 *                          gain, off (DESIGN.md sections 3.2c, 4)
 *   "ext_erode"        (2) extended algorithm, the erosion kernel: 2 / 1 = a wave per 62 word columns of the bit plane and 16 / 32
 *                          rows (one load per row and lane, all in flight, neighbours' words by DPP, XCD-aware block map), 0 = round 1's
 *                          kernel (a lane per word column, three loads per row); "ext_e_sparse" (0): 1 = the signal-region plane is
 *                          cleared behind the previous batch and only its non-zero words are stored (measured: the fill costs the
 *                          first pass more than the stores cost the erosion)
 * The reference has no counterpart (its launch wrapper has one path, spotfinder/spotfinder.cu:148-189). */
int ffs_ctx_set_tuning(ffs_ctx *ctx, const char *key, long long value);

/* ---- streams: one in-flight batch each (replaces one worker thread's CudaStream +
 *      pinned/device buffers, spotfinder.cc:729-742) --------------------------------------------
 * An ffs_stream owns its buffers and its batch; the HIP streams underneath belong to the context (one for
 * the dense kernels of all its ffs_streams, in submission order, two for the sparse launches, one for uploads:
 * DESIGN.md section 3.4).  Distinct ffs_streams may be driven from distinct threads at the same time; keep
 * two or more batches in flight (two or more ffs_streams) for throughput -- a batch's sparse stage runs beside
 * the next batch's threshold kernel. */
int ffs_stream_create(ffs_ctx *ctx, ffs_stream **out);
/* Waits for the stream's batch in flight, if any, and frees its buffers (idempotent: see ffs_ctx_destroy). */
void ffs_stream_destroy(ffs_stream *s);

/* The stream's pinned staging area (max_batch dense frames): decode straight into it, as the
 * reference's workers decompress into their pinned host_image (spotfinder.cc:731, :828-842),
 * then pass the same pointer to ffs_submit(). */
int ffs_stream_host_buffer(ffs_stream *s, void **ptr, size_t *bytes);
/* The staging area is pinned memory, which costs ~170 ms per GB to allocate (and the runtime serialises the
 * allocations of all threads): it is allocated on first use, max_batch raw frames by default.  A producer of
 * compressed chunks that knows it needs less (max_batch chunks, not max_batch frames) says so BEFORE its first
 * ffs_stream_host_buffer(); a later call with a larger size grows the area (contents are not kept; not while a
 * batch is in flight).  Staging areas of destroyed streams are reused by the context's next streams. */
int ffs_stream_reserve_host(ffs_stream *s, size_t bytes);

/* cudaMemcpy2DAsync H2D + call_do_spotfinding_dispersion + D2H + ConnectedComponents
 * (spotfinder.cc:846-905), for n_frames dense host frames (each W*H pixels, consecutive).
 * Asynchronous; results are collected with ffs_wait(). */
int ffs_submit(ffs_stream *s, const void *host_pixels, uint32_t n_frames,
               int64_t first_frame_id);
/* Same, frames already in device memory (rows pitch_bytes apart, frames frame_stride_bytes
 * apart; pitch_bytes a multiple of 16).  For producers that decode on the GPU, and for bench.py's
 * resident-in-HBM measurement.  The pixels must stay valid and unchanged until ffs_wait() has returned
 * (a frame that exceeds the stream's list capacity is read a second time there). */
int ffs_submit_device(ffs_stream *s, const void *device_pixels, size_t pitch_bytes,
                      size_t frame_stride_bytes, uint32_t n_frames, int64_t first_frame_id);
/* Device layout this context prefers for ffs_submit_device (and uses internally). */
int ffs_ctx_device_layout(const ffs_ctx *ctx, size_t *pitch_bytes, size_t *frame_stride_bytes);

/* Blocks until the stream's batch is done; results[i] describes frame i of the batch. */
/* Same as ffs_submit, but each frame arrives as the raw bitshuffle-LZ4 chunk the detector wrote
 * (12-byte header + blocks: what Reader::get_raw_chunk returns, h5read.c:428-456, shmread.cc) and is
 * decoded on the GPU -- replaces bshuf_decompress_lz4 on the worker thread (spotfinder.cc:823-842,
 * h5read/src/read_chunks.cc:22-24) and moves 4-6x fewer bytes over PCIe.  chunks[i] may point
 * anywhere (copied into the stream's pinned staging buffer) or, for zero copy, inside the buffer
 * ffs_stream_host_buffer returns (then all of them must).  The call starts the PCIe copy and returns;
 * a helper thread of the stream indexes the blocks and enqueues the kernels meanwhile (ffs_wait
 * joins it).  A chunk whose header does not say width*height*pixel_bytes is refused here
 * (FFS_ERR_INVALID); block lengths that run past chunk_bytes[i] and corrupt LZ4 streams are
 * reported by ffs_wait (FFS_ERR_INVALID). */
int ffs_submit_compressed(ffs_stream *s, const void *const *chunks, const size_t *chunk_bytes,
                          uint32_t n_frames, int64_t first_frame_id);
/* Decode only, for tests and benchmarks: n_frames chunks -> the stream's device image buffer
 * (default layout); the average duration of one decode launch in ms_decode (HIP events, iters
 * launches).  host_out (may be NULL) receives the decoded frames, W*H*pixel_bytes each. */
int ffs_decode_only(ffs_stream *s, const void *const *chunks, const size_t *chunk_bytes,
                    uint32_t n_frames, uint32_t iters, float *ms_decode, void *host_out);

/* (ffs_wait turns the batch's records into the arrays below; for large batches the frames are spread over the calling thread and
 * three helper threads that belong to the context -- created on first use, joined by ffs_ctx_destroy.) */
int ffs_wait(ffs_stream *s, const ffs_frame_result **results, uint32_t *n_results);

/* The whole batch's boxes and reflections as two contiguous arrays (frame i's slice starts
 * where frame i-1's ends; lengths are results[i].n_boxes / .n_reflections) -- lets a binding
 * wrap a batch without touching every frame. */
int ffs_stream_batch_arrays(ffs_stream *s, const ffs_box **boxes, uint32_t *n_boxes,
                            const ffs_reflection **reflections, uint32_t *n_reflections);

/* Timings of the last completed batch on this stream, milliseconds (HIP events on the
 * stream): [0] H2D, [1] threshold kernels, [2] compaction + connected components,
 * [3] D2H, [4] total.  (The reference prints Copy/Kernel/Post Copy/Post, spotfinder.cc:1056-1076.) */
int ffs_stream_timings(ffs_stream *s, float ms[5]);

/* Which launches the stream's last batch took -- for tests and tools that must know that the path they mean to exercise ran
 * (results never depend on it): a mask of the bits below; *reruns = how many times ffs_wait ran the batch again because a plan
 * did not hold it (list / component capacity, LDS forests, wave logs, band plan).  Either pointer may be NULL.
 * (The reference has one path: spotfinder/spotfinder.cu:148-189 + the host's ConnectedComponents.) */
#define FFS_PATH_WAVE_LOGS 1u      /* the streaming kernel's strong groups travelled in per-wave logs, not the bit plane */
#define FFS_PATH_FRAME_CHAIN 2u    /* sparse stage: one launch, a workgroup per frame */
#define FFS_PATH_BANDS 4u          /* sparse stage: a wave per band of a frame + a merge per frame */
#define FFS_PATH_RUNS 8u           /* ... the one launch's forest over runs of strong pixels (dense frames) */
#define FFS_PATH_GRID_KERNELS 16u  /* sparse stage: four grid-wide kernels */
#define FFS_PATH_EXTENDED 32u      /* extended dispersion */
int ffs_stream_last_path(ffs_stream *s, uint32_t *path_bits, uint32_t *reruns);

/* Centres of mass of the last batch's reflections as rows (frame_id, x, y, z) of float32 -- the
 * payload of `--output-for-index` (spot_centers, spotfinder.cc:919-933,1004-1006), in frame order;
 * used to feed a multi-GPU gather without a per-frame loop on the caller's side.  Lane 0 carries the
 * low 32 bits of the frame id as a BIT PATTERN (read it as uint32: as a float value ids would collide
 * from 2^24 on).  Writes at most `cap` rows, then one more row whose first two lanes are, again as uint32
 * bit patterns, (rows written, rows wanted): rows4 must hold (cap + 1) * 4 floats.  Returns
 * FFS_ERR_OVERFLOW (rows written are valid) when wanted > cap.  Needs want_reflections.  Reads the last ffs_wait's results
 * only: may run (on another thread) while the stream's next batch is in flight, until the next ffs_wait on this stream. */
int ffs_stream_spot_centres(ffs_stream *s, float *rows4, uint32_t cap, uint32_t *n_written);

/* ---- measurement entry points (bench.py roofline leg, tools/) -------------------------------- */
/* Runs the threshold stage alone on device-resident frames, `iters` times, every launch on the state the hot
 * path gives it (zeroed counters, empty plane) and returns the average duration of ONE launch of its dense
 * kernel (ms_dense: the streaming kernel; extended algorithm: its first pass) and of what follows it inside the
 * stage (ms_rest: the bright-window fix-up; extended: erosion + final pass), each from HIP events that ride on
 * the dispatches themselves, on the stream the kernels are launched on.  (The reference prints "Kernel: ms" per
 * image from events around its launch wrapper, spotfinder/spotfinder.cc:1056-1076.) */
int ffs_bench_threshold(ffs_stream *s, const void *device_pixels, size_t pitch_bytes,
                        size_t frame_stride_bytes, uint32_t n_frames, uint32_t iters,
                        float *ms_dense, float *ms_rest);
/* The submit / wait loop of a resident-frames benchmark, natively: `steps` batches of the same device-resident
 * frames through `n_streams` streams of ONE context, all of them in flight (ffs_submit_device / ffs_wait).  For
 * drivers with one host thread per GPU (bench.py --single-process, the threading model of
 * spotfinder/spotfinder.cc:725-752 spread over several devices).  Returns the sums over all frames. */
int ffs_bench_pipeline(ffs_stream *const *streams, uint32_t n_streams, const void *device_pixels,
                       size_t pitch_bytes, size_t frame_stride_bytes, uint32_t n_frames, uint32_t steps,
                       int64_t first_frame_id, uint64_t *n_boxes, uint64_t *n_strong_pixels);
/* Memory ceiling measured on this device (BASELINE.md section 3 asks for one beside the nominal 8 TB/s): the
 * stream's own buffers are read linearly, 16 B per lane (read_gbps), and read while one 8-byte zero store per
 * 16 bytes read goes to the byte-mask buffer -- the threshold kernel's 2:1 read/write mix with the friendliest
 * possible addresses (mix_gbps).  GB/s of bytes moved.  Overwrites the stream's byte masks; no batch in flight.
 * (The reference prints GBps per image, spotfinder/spotfinder.cc:1056-1076, against no ceiling.) */
int ffs_bench_hbm(ffs_stream *s, uint32_t iters, float *read_gbps, float *mix_gbps);
/* Device pointers of the last batch's dense planes (strong byte mask rows are
 * mask_pitch apart) -- for parity tests that want the raw kernel output.  The byte masks hold the last batch's
 * result only if that batch ran with want_strong_mask = 1. */
int ffs_stream_debug_planes(ffs_stream *s, const uint8_t **device_strong_bytes,
                            size_t *mask_pitch, size_t *mask_frame_stride);
/* Copies one frame's bit plane of the last completed batch to host memory, unpacked to W*H
 * bytes (0/1): which = 0 strong pixels, 1 the extended algorithm's first-pass "not background"
 * mask (first_pass_dispersion_result in the reference's --writeout, spotfinder.cu:268-279),
 * 2 its eroded signal region (eroded_dispersion_result, :300-311). */
int ffs_stream_debug_bitplane(ffs_stream *s, uint32_t frame_in_batch, int which, uint8_t *host_out);

/* Known-answer self test of the one non-trivial fp64 operation the exact predicate relies on:
 * sum (mod 2^64) of the bit patterns of sqrt((double)n) for integers n in [begin, end), computed on
 * the device with the same code path k_exact uses.  The caller compares it with libm's correctly
 * rounded sqrt (tests/test_gpu_numerics.py does, exhaustively for every n = x*m a uint16 frame
 * can produce). */
int ffs_selftest_sqrt(ffs_ctx *ctx, uint64_t begin, uint64_t end, uint64_t *sum_of_bits);

/* ---- rotation sweeps: 3D connected components (replaces
 *      ConnectedComponents::find_3d_components, connected_components.cc:270-470) -------------- */
int ffs_stack3d_create(ffs_ctx *ctx, uint64_t max_total_strong, ffs_stack3d **out);
/* (idempotent: see ffs_ctx_destroy; the stack's buffers are kept by the context for its next sweep) */
void ffs_stack3d_destroy(ffs_stack3d *st);
/* Adds the strong pixels of every frame of the stream's last completed batch, keyed by frame_id
 * (the reference keys its rotation_slices map by image number, spotfinder.cc:913-918). */
int ffs_stack3d_add_batch(ffs_stack3d *st, ffs_stream *s);
/* Adds one slice from host memory (k ascending) -- what a gather from other GPUs delivers. */
int ffs_stack3d_add_slice(ffs_stack3d *st, int64_t frame_id, const uint32_t *k,
                          const uint32_t *intensity, uint32_t n);
/* Orders slices by frame_id (std::map order, spotfinder.cc:1105-1108), z = rank, runs the 3D
 * union-find on the device, filters with min_spot_size_3d / max_peak_centroid_separation.
 * reflections are in label order; n_calculated = "Calculated {} spots". */
int ffs_stack3d_finish(ffs_stack3d *st, const ffs_reflection **reflections,
                       uint32_t *n_reflections, uint32_t *n_calculated,
                       uint32_t *n_filtered_size, uint32_t *n_filtered_sep);
/* ---- several GPUs in one process ------------------------------------------------------------------
 * The reference drives ONE device (-d, src/ffs/cuda_arg_parser.cc:30-61).  A driver that owns several makes
 * one ffs_ctx per GPU and deals its frame queue to all of them (spotfinder --devices / --gpus); stills need
 * no exchange.  For rotation sweeps ffs_stack3d_add_batch() accepts a stream of ANOTHER context with the same
 * frame shape: the batch's strong-pixel lists are packed on their GPU and sent to the stack's GPU -- by RCCL
 * point-to-point over xGMI when ffs_multi_init() could load librccl and build the communicators, else by
 * peer copies.  `devices`: the GPUs in use (duplicates allowed); transport: "rccl", "peer" or NULL (= the
 * FFS_GATHER environment variable, default "rccl").  ffs_multi_transport(): "rccl", "peer" or "none".
 * Call it before the first batch is added to a stack, not while batches are being added.  A frame that
 * overflowed its stream's lists reaches the stack from any GPU (its list goes up from the host, as on one GPU). */
int ffs_multi_init(const int *devices, int n_devices, const char *transport);
const char *ffs_multi_transport(void);
/* The gather of the per-frame spot lists over RCCL (BASELINE.json north_star, SURVEY section 8(e): "ncclAllGather of per-rank record
 * counts, then grouped ncclSend / ncclRecv of the packed spot records to rank 0, then a single D2H"; the reference has one device and
 * no such step).  streams[i]: one stream per taking-part context, each with a completed batch (after ffs_wait) whose centre rows
 * (frame id bits, x, y, z: ffs_stream_spot_centres) are what travels; every rank of the communicator ffs_multi_init built needs at
 * least one stream, contexts that share a GPU share its rank (their rows travel as one message).  The rows arrive in rows4_out (host
 * memory, room for cap rows of four floats) rank after rank, inside a rank in the order of `streams`; *n_rows = rows wanted.  FFS_ERR_OVERFLOW: they do not fit cap; FFS_ERR_INVALID: no RCCL
 * communicators (ffs_multi_transport() != "rccl").  A stream of the root's own GPU hands its rows over by a device copy unless the
 * transport was forced (ffs_multi_init(.., "rccl") or FFS_GATHER): then it sends to its own rank -- what a one-GPU box can rehearse.
 * In ONE process the rows are on the host already after ffs_wait; reading them there is the faster gather:
 * this entry point is the A/B partner and the building block for drivers that keep the rows on the devices. */
int ffs_multi_gather_rows(ffs_stream *const *streams, uint32_t n_streams, uint32_t root, float *rows4_out, uint32_t cap,
                          uint32_t *n_rows);
/* NUMA node the GPU hangs off (sysfs), -1 if unknown: where a driver should keep the worker threads that feed
 * it (the reference pins nothing: one device, spotfinder.cc:725-742). */
int ffs_device_numa_node(int device);
/* Device time of the last ffs_stack3d_finish (upload of the slice table to the labels of every strong pixel), ms. */
int ffs_stack3d_last_finish_ms(const ffs_stack3d *st, float *ms);
/* Per-signal view of the last ffs_stack3d_finish: every strong pixel of the stack in the reference's
 * vertex order (slice by slice, ascending linear index -- the order Reflection3D::signals_ is filled
 * in, connected_components.cc:409-446) with the index of its reflection in the array finish
 * returned, or -1 if that spot was filtered.  For per-signal sums the caller owns, e.g.
 * Reflection3D::variances_in_kabsch_space (connected_components.cc:159-203).  Pointers stay valid
 * until the next finish / destroy. */
int ffs_stack3d_signals(ffs_stack3d *st, const uint32_t **x, const uint32_t **y, const int32_t **z,
                        const uint32_t **intensity, const int32_t **reflection, uint64_t *n);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif
