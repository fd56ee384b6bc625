// ffs_device.h -- shared host/device definitions for libffs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing experiments (phases switched off, kernels stopped half way: results are wrong when used) exist only in builds
// with -DFFS_EXPERIMENTS (make experiments -> libffs_hip_exp.so, for tools/); the product library has none of them.
#ifdef FFS_EXPERIMENTS
#define FFS_DBG(args, bit) (((args).dbg & (bit)) != 0)
#define FFS_STOP_AFTER(A, phase) if ((A).stop_after == (phase)) return
// (experiments) device timestamps at the phase boundaries of the sparse launch, one set per frame: tools/chain_phase_times.sh
#define FFS_PHASE_TS(A, idx) do { if ((A).phase_ts && threadIdx.x == 0) (A).phase_ts[(size_t)blockIdx.x * 8 + (idx)] = wall_clock64(); } while (0)
#else
#define FFS_PHASE_TS(A, idx) do {} while (0)
#define FFS_DBG(args, bit) false
#define FFS_STOP_AFTER(A, phase) do {} while (0)
#endif

namespace ffsamd {

// ---- frame layout in HBM ---------------------------------------------------------------------
// Pixels:      [frame][y][pitch_px] of PixelT, rows `pitch` bytes apart (multiple of 128 B).
// Bit planes:  [frame][y][mpitch] bytes, 1 bit per pixel, LSB first (bit x&7 of byte x>>3);
//              bits for x >= W are always 0.  Used for: valid-pixel mask (one plane per
//              context, frame-invariant), candidate plane, strong plane (candidate plane
//              filtered in place).
// Byte mask:   [frame][y][bpitch] bytes 0/1 -- the reference kernel's result_strong layout
//              (spotfinder/kernels/thresholding.cu:233), kept as the drop-in contract.
struct Layout {
    int W, H;
    int pitch_px;        // multiple of 64, >= W
    uint32_t pitch;      // bytes per pixel row (default layout)
    uint32_t mpitch;     // bytes per bit-plane row  = pitch_px / 8
    uint32_t bpitch;     // bytes per byte-mask row  = pitch_px
    uint64_t frame_stride;       // bytes between frames (default layout)
    uint64_t plane_frame_stride; // bytes between frames of a bit plane = H * mpitch
    uint64_t bytes_frame_stride; // bytes between frames of the byte mask = H * bpitch
};

// ---- streaming threshold kernels: geometry (kernels_stream.hpp) ---------------------------------------
// One wave64 marches down a column strip; a lane holds 16 bytes of a pixel row (8 pixels of 16 bits, 4 of 32).
// Lanes 0 and 63 are halo (their windows are incomplete), lanes 1..62 own output.
constexpr int kSOwned = 62;
// first row of band b for a (split, rows, rows2) geometry -- see ThresholdArgs::band_split
__host__ __device__ inline int band_first_row(int band, int band_rows, int band_rows2, int band_split) {
    return band < band_split ? band * band_rows : band_split * band_rows + (band - band_split) * band_rows2;
}
constexpr int kInfoExtraRows = 3;    // ginfo row y carries the mask bits of row y and the window counts of row y - 3

// Exact-stage tiles: one 256-thread workgroup per 8 rows.
constexpr int kWlogCap = 256;        // entries of a wave's log (a wave of the bench frames writes ~40)
constexpr int kTileRows = 8;
constexpr int kExactListCap = 2048;  // candidate entries staged in LDS per flush

struct ThresholdArgs {
    const void* image;         // device pixels
    uint64_t frame_stride;     // bytes
    uint32_t pitch;            // bytes
    const uint8_t* maskbits;   // valid-pixel bit plane [H][mpitch]
    uint8_t* bits;             // candidate / strong bit planes [n][H][mpitch]
    uint8_t* strong_bytes;     // byte masks [n][H][bpitch]
    uint32_t* tile_counts;     // [n][n_tiles] strong pixels per exact-stage tile
    int W, H, pitch_px;
    uint32_t mpitch, bpitch;
    uint64_t plane_frame_stride, bytes_frame_stride;
    int n_strips, band_rows, n_bands, n_tiles;
    // Bands of two heights (tuning "band_taper"): bands 0 .. band_split - 1 are band_rows tall, the rest band_rows2.  The block map
    // hands out bands in ascending order, so with band_rows2 < band_rows the LAST waves of a launch are the short ones and the
    // launch's tail (the slots of the machine that idle while the last round of waves finishes) shrinks.  band_split = n_bands: uniform.
    int band_rows2, band_split;
    // parameters
    float kS;                  // nsig_s^2 (1 - 2^-16): conservative signal pre-filter
    float kB;                  // nsig_b (1 - 2^-20): conservative dispersion pre-filter
    int min_count;
    double nsig_b, nsig_s, threshold;
    double nsig_b2, nsig_s2;   // squares (float64), for the square-root-free form of the predicate
    long long max_valid;       // < 0: no test
    // integer form of the predicate (kernels_stream.hpp: int_predicate): usable when nsig_b^2 and nsig_s^2 are integers
    int int_pred;              // 1: ib2 / is2 / thr_floor are valid
    uint32_t ib2, is2;         // nsig_b^2, nsig_s^2
    uint32_t thr_floor;        // floor(threshold): for an integer pixel p, p > threshold <=> p > floor(threshold)
    int bright_to_plane;       // streaming kernels: 0 = bright windows go onto bright_list (k_bright_fix decides them);
                               // 1 = they are marked in the plane as candidates and the exact kernel filters the plane;
                               // 2 (host side only) = the plane starts as the valid-pixel mask and the exact kernel decides every pixel
    // extended dispersion (kernels_extended.hpp)
    uint8_t* dplane;           // first-pass "not background" bit planes [n][H][mpitch]
    uint8_t* eplane;           // eroded signal-region bit planes [n][H][mpitch]
    int eplane_clean;          // the signal-region plane is all zero (cleared behind the previous batch): the erosion stores its non-zero words only
    int ext_strips, ext_band_rows, ext_bands;
    int ext_flavour;           // 0 = baseline.cpp rules, 1 = device-kernel rules
    int ext_variant;           // first pass, 16-bit pixels: 2 = streaming kernel (k_stream_u16<true>), 0 = k_ext_first
    // one-kernel threshold for 16-bit pixels (kernels_stream.hpp)
    const uint8_t* ginfo;      // [H + 3][gpitch] one dword per 8-pixel group: mask bits | min count << 8 | max count << 16
    const uint8_t* mmap;       // [H][pitch_px] 7x7 window count of every pixel
    uint32_t gpitch;           // bytes per ginfo row = pitch_px / 2
    int gpf;                   // lane groups per frame row = ceil(W / 8)
    int n_frames;              // frames of this launch
    int group_frames;          // frames laid side by side in one super row (group bytes < 2 GiB)
    int s_strips, s_band_rows, s_bands;
    uint32_t* bright_n;        // [1] pixels handed to k_bright_fix (window sum >= 65536)
    uint2* bright_list;        // [bright_cap] (frame << 16 | x, y)
    uint32_t bright_cap;
    uint32_t* overflow;        // status word: 8 = the bright-window list overflowed
    // occupancy of the strong plane, one bit per 16-byte segment (128 pixels) of a plane row: set by whoever sets a plane
    // bit in a one-kernel batch, read and cleared by k_frame_chain, which then loads only the segments that hold something
    uint32_t* occ;             // [n][occ_frame_words]
    uint32_t occ_frame_words;  // words per frame = ceil(H * occ_spr / 32)
    uint32_t occ_spr;          // segments per plane row = mpitch / 16
    int dense_mask;            // the streaming kernels zero-fill the byte mask (somebody wants it); else they leave it alone
    // wave logs (tuning "strong_log", 16-bit standard path): instead of scattering plane bytes, counter and occupancy atomics
    // and bright-list entries over the stream's buffers, every wave appends one 8-byte entry per group that holds a strong
    // (or undecided) pixel to its own log -- dense stores; kernels_chain.hpp merges the logs of a frame
    uint2* wlog;               // [waves][kWlogCap] (row << 16 | group, frame in super row << 16 | undecided << 8 | strong); nullptr: off
    uint32_t* wlog_n;          // [waves] entries a wave wanted to write (more than kWlogCap: overflow)
    uint4* wpix;               // [waves][kWlogCap] the entry's eight (masked) pixels: the sparse launch never gathers from the image
    // hand-over to the next streaming kernel (ffs_submit.hip, tuning "dense_overlap"): the launch's LAST workgroup -- dispatch is in
    // order -- writes handoff_seq here as it starts; the other dense HIP stream waits for that value before ITS kernel, which then
    // flows into the slots this launch's last round of waves leaves.  nullptr: off.
    uint32_t* handoff;
    uint32_t handoff_seq;
    int dbg_prio;              // tuning "stream_prio": the 16-bit streaming kernel raises its waves' issue priority (s_setprio 3)
    int dbg;                   // timing experiments (-DFFS_EXPERIMENTS builds only; results are wrong when set)
};

// The streaming launch's units.  A unit = one wave = one band of one strip; units are numbered band after band (u = band * n_strips +
// strip) and dealt to the eight XCDs in eight contiguous chunks (workgroup b runs on XCD b % 8): the strips of a band -- neighbours that
// share halo columns and, frame after frame, the same rows of the mask tables -- share an L2, and ANY number of bands is balanced over
// the XCDs.  (Rounds 1-5 dealt the bands round-robin, band = xcd + 8 k, which wanted a multiple of eight bands; the chunks measure
// 0.5-1 % faster on the same box, profiles/r06e_map_variants.log.)
__host__ __device__ inline uint32_t stream_units(const ThresholdArgs& a) { return (uint32_t)a.n_bands * (uint32_t)a.n_strips; }
__host__ __device__ inline uint32_t stream_chunk(const ThresholdArgs& a) { return (stream_units(a) + 7u) / 8u; }   // units per XCD; grid.x = 8 chunks
__device__ __forceinline__ bool stream_unit(const ThresholdArgs& a, uint32_t bid, int& strip, int& band) {
    const uint32_t u = (bid & 7u) * stream_chunk(a) + (bid >> 3);
    band = (int)(u / (uint32_t)a.n_strips);
    strip = (int)(u - (uint32_t)band * (uint32_t)a.n_strips);
    return u < stream_units(a);
}
// where the log of (super row y, band, strip) lies in wlog / wlog_n / wpix: the strips of a band side by side
__host__ __device__ inline uint32_t log_slot(const ThresholdArgs& a, uint32_t y, uint32_t band, uint32_t strip) {
    return (y * (uint32_t)a.n_bands + band) * (uint32_t)a.n_strips + strip;
}

// ---- strong-pixel lists and connected components -------------------------------------------------
// 2D only (round 2): the accumulator of a component sits at the list index of its ROOT (its smallest
// member, always the first pixel of a horizontal run), so no pass has to number the components before
// they can be reduced.  k_emit_list_w resets the accumulator of every run start.
struct CompAcc2 {
    unsigned long long sum_i, sum_xi, sum_yi, peak;
    uint32_t x_min, x_max, y_min, y_max;
    uint32_t num_pixels, pad;
};

// 2D record as it crosses PCIe (40 bytes instead of the 56 of ffs_reflection: the z fields are constants
// for a single frame and the coordinates fit 16 bits); ffs_wait() widens it again.
struct WireRec2 {
    uint16_t x_min, x_max, y_min, y_max;
    uint32_t npx_flags;        // num_pixels | flags << 30
    float com_x, com_y;
    uint16_t peak_x, peak_y;
    uint32_t peak_intensity;
    float peak_centroid_distance;
    unsigned long long sum_intensity;
};

struct CclArgs {
    const void* image;
    uint64_t frame_stride;
    uint32_t pitch;
    uint8_t* bits;             // strong bit planes
    int clear_bits;            // clear every plane word after reading it (k_stream_u16 needs an all-zero plane)
    const uint32_t* tile_counts;
    uint32_t* num_strong;      // [n]
    uint32_t* row_off;         // [n][H+1] list offset of the first strong pixel of each image row
    uint32_t* list_k;          // [n][cap] ascending linear index y*W + x
    uint32_t* list_i;          // [n][cap] intensity
    uint32_t* parent;          // [n][cap]
    uint32_t* comp_id;         // [n][cap] component number of each ROOT entry
    uint32_t* n_comp;          // [n]
    uint32_t* overflow;        // [1] set if any frame exceeded cap / max_comp
    int W, H, pitch_px;
    uint32_t mpitch;
    uint64_t plane_frame_stride;
    int n_tiles;
    uint32_t cap;              // list capacity per frame
    uint32_t max_comp;         // record capacity per frame
    int pixel_bytes;
    CompAcc2* acc2;            // [n][cap] or null: reset the accumulator of every run start (k_reduce_roots)
    uint32_t* summary;         // [n][8]: with acc2, the compaction zeroes n_comp and summary (the root-indexed kernels add into them)
    uint8_t* strong_bytes;     // byte masks [n][H][bpitch]: the compaction sets the 1s (the threshold kernels zero-fill)
    uint32_t bpitch;
    uint64_t bytes_frame_stride;
    int dense_bytes;           // write the byte mask at all (only when somebody asked for it)
    int need_lists;            // the strong-pixel lists (k, intensity) have a reader after the launch (host copy, 3D stack); 0: the launch that
                               // merges wave logs keeps them to itself (1.15 M scattered 4-byte stores per batch less)
    uint32_t* occ;             // see ThresholdArgs; null or use_occ == 0: the whole plane is read
    uint32_t occ_frame_words, occ_spr;
    int use_occ;
};

// Per-component accumulator (device) -- reduced with 64-bit integer atomics so the result
// is independent of arrival order (the reference's double sums of half-integer * integer
// terms are exact, connected_components.hpp:86-90, so integer sums reproduce them bit for bit).
struct CompAcc {
    unsigned long long sum_i;    // sum I
    unsigned long long sum_xi;   // sum (2x+1) I
    unsigned long long sum_yi;   // sum (2y+1) I
    unsigned long long sum_zi;   // sum (2z+1) I
    unsigned long long peak;     // (I << 32) | ~(position rank): max => highest I, then smallest (z,y,x)
    uint32_t x_min, x_max, y_min, y_max;
    int32_t z_min, z_max;
    uint32_t num_pixels;
    uint32_t root;               // list index of the minimum vertex (label order key)
};

// A segment is one independent labelling problem: a frame (2D) or a whole z-stack (3D).
struct SegArgs {
    const uint32_t* list_k;   // per segment: ascending (z, k)
    const uint32_t* list_i;
    uint32_t* parent;
    uint32_t* comp_id;        // 3D only: component number of each root entry
    const uint32_t* seg_n;    // [n_seg] entries per segment
    uint64_t seg_stride;      // entries between segment starts
    uint32_t* n_comp;         // [n_seg]
    CompAcc* acc;             // [n_seg][max_comp]
    uint32_t max_comp;
    uint32_t* overflow;
    uint32_t W, H;
    const uint32_t* row_off;  // 2D only: [n_seg][H+1] per-row list offsets (may be null)
    // 3D only
    const uint32_t* slice_begin;  // [n_slices + 1] entry offsets of each slice inside the segment
    int n_slices;
    const uint32_t* zs;           // slice of every entry (k_stack_gather), or null
    // finalize
    uint32_t min_spot_size;
    float max_sep;
    void* recs;               // ffs_reflection, packed: segment s starts at sum_{q<s} n_comp[q]
    uint32_t* summary;        // [n_seg][8]: n_boxes, n_strong_filtered, n_refl, n_filt_size, n_filt_sep
    // 2D, accumulators at the root (k_reduce_roots / k_finalize_roots)
    CompAcc2* acc2;           // [n_seg][seg_stride]
    uint32_t* chunk_roots;    // [n_seg][chunks_max] roots per chunk of kRootChunk list entries
    uint32_t chunks_max;
    // 2D: the compaction has consumed the per-tile counts; k_union zeroes them (and the bright-list count) again, which
    // is what the next batch's streaming kernel adds into
    uint32_t* zero_counts;    // [n_seg][zero_per_seg], or null
    uint32_t zero_per_seg;
    uint32_t* zero_word;      // one more word to clear, or null
};


// One slice of the 3D stack's copy tables (kernels_stack3d.hpp)
struct StackSlice {
    uint32_t src;   // offset in the arrival buffers (k_stack_gather) / in the stream's list of that frame (k_stack_append)
    uint32_t dst;   // offset in the destination buffers
    uint32_t n;     // entries
    uint32_t z;     // k_stack_gather: position of the slice in the stack
};

// Matches ffs_reflection in include/ffs_hip.h
struct ReflOut {
    uint32_t x_min, x_max, y_min, y_max;
    int32_t z_min, z_max;
    int32_t num_pixels;
    float com_x, com_y, com_z;
    uint32_t peak_x, peak_y;
    int32_t peak_z;
    uint32_t peak_intensity;
    float peak_centroid_distance;
    uint32_t flags;
    unsigned long long sum_intensity;
};


}  // namespace ffsamd
