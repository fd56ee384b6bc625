// kernels_chain.hpp (included by ffs_submit.hip) -- the whole sparse stage of a frame in ONE workgroup.
//
// Why: the sparse stage (compaction -> union-find -> reduction -> records) is a chain of dependent memory round
// trips over ~10^4 strong pixels per frame.  As four grid-wide kernels it cost the batch their full durations
// (~130 us of a 500 us step) although they keep a few per cent of the machine busy: a kernel that becomes ready
// while another queue's 15 000-workgroup streaming kernel is being dispatched gets no CU until that dispatch has
// drained, so each of the four waited for a tail of its own.  ONE launch per batch needs
// one tail, and then runs beside the next batches' streaming kernels for as long as it likes: what it costs the
// machine is the slots it holds (one workgroup per frame), not its duration.
//
// With a whole frame in one workgroup the round trips can stay on the CU.  A dependent access costs ~1 us at the
// L2 and ~0.1 us in LDS, so the frame's union-find forest, the per-row list offsets and the list's column numbers
// (16 bits each: with the row offsets they are the list) live in LDS (frames up to kChainLdsEntries strong pixels;
// denser frames run the stages on the global arrays), and the accumulators of 832 components at a time do too.
//
// Phases (separated by __syncthreads(), which also orders the block's global writes):
//   B  (only while few batches are in flight, see ffs_submit.hip) the bright-window fix-up, k_bright_fix's work, for the
//      listed pixels of this frame; the last workgroup through with the list empties it
//   A  exclusive scan of the frame's per-tile counts (the streaming kernel's atomics) -> tile offsets in LDS; the
//      counts are zeroed for the next batch
//   E  compaction: each wave takes a contiguous range of tiles; it reads the occupancy bitmap the streaming kernel
//      left (one bit per 16-byte plane segment), lists the occupied segments in order and loads only those -- or,
//      without a bitmap (extended algorithm), streams its whole range (buffer loads, twelve rounds of 256 words in
//      flight, tiles without strong pixels skipped); the non-zero words are staged in LDS and their pixels placed
//      64 words at a time; nothing here waits for a load it just issued (the pixel VALUES are fetched in phase P)
//   S  per-row counts -> per-row list offsets (block scan, in LDS)
//   X  the list's column numbers into LDS
//   U  union-find: vertical edges + the reference's row-wrap edge (k_union<false>'s edges), all in LDS
//   then, for frames held in LDS:
//   P  pixel values (twenty independent loads per thread); every entry finds its root; roots are numbered in list order (= label order, connected_components.cc:91,242)
//   R  832 components at a time: entries add into LDS accumulators (integer atomics: order-independent), one thread
//      per component writes its 40-byte record, the records leave as consecutive dwords
//   and for denser frames: R' accumulators at the root's list index in global memory, F' records chunk by chunk
//   (the bodies of k_reduce_roots / k_finalize_roots).
// Three instantiations (template parameters RUNS, LOG):
//   plain        the phases above: the streaming kernels' (or the extended algorithm's) bit plane, forest over pixels
//   RUNS         frames of 16-bit pixels beyond the LDS forest of pixels: phases E' X' U' P' R' over RUNS of strong pixels (a
//                run = a maximal row of strong pixels inside one 32-pixel plane word: 61 k pixels are 12 k runs on the extended
//                algorithm's bench frames)
//   LOG          the standard path (round 3c): no plane at all -- the streaming kernel appends its strong groups, with their
//                pixels, to per-wave logs (ThresholdArgs::wlog / wpix); phases L1 (per-row counts; undecided bright-window
//                pixels listed and decided one per thread) and, after S, L2 (the logs of the strips of every band merged into
//                the raster-order list, column numbers into LDS, intensity list from the logged pixels) replace B, A, E, X and
//                the image gathers of P
// Frame f's records start at f * max_comp of the (host) record buffer -- no frame waits for another's count -- and
// counters and flags go straight into the pinned block ffs_wait() reads: no copy follows the kernel.
// Results are those of the four-kernel chain bit for bit (same edges, same integer accumulators, same order).
#pragma once
#include "kernels_ccl.hpp"
#include "kernels_threshold.hpp"   // exact_strong: the bright-window fix-up runs here when the batch has one launch only

namespace ffsamd {

constexpr int kChainThreads = 1024;
constexpr int kChainWaves = kChainThreads / 64;
constexpr int kChainMaxRows = 4480;     // per-row offsets kept in LDS; taller frames take the four kernels
constexpr int kChainMaxTiles = kChainMaxRows / kTileRows;
constexpr int kChainListCap = 320;      // non-zero plane words a wave stages (a round of loads adds at most 256)
constexpr int kChainSegCap = 512;       // occupied plane segments a wave lists per round of its occupancy bitmap (8 bits per lane)
constexpr int kChainQuads = 4;          // rounds (4 words per lane each) per batch of plane loads; three batches are live
constexpr int kChainLdsEntries = 20480; // strong pixels of a frame whose union-find forest fits LDS
constexpr int kChainPer = kChainLdsEntries / kChainThreads;   // consecutive entries per thread in phases U / P / R
constexpr int kChainSlots = 832;        // components accumulated in LDS at a time
// Denser frames of 16-bit pixels (the extended algorithm marks whole spots: 61 k strong pixels in 12 k runs on the bench
// frames): the forest is built over RUNS -- maximal rows of strong pixels inside one 32-pixel plane word -- instead of pixels.
constexpr int kChainRunCap = 16384;     // runs of a frame whose forest (4 B) and descriptors (4 B) fit LDS
constexpr int kChainRunPer = kChainRunCap / kChainThreads;   // consecutive runs per thread in phases U' / P' / R'
constexpr int kChainRunMaxW = 16383;    // descriptor: y << 19 | x0 << 5 | (len - 1)  (H <= kChainMaxRows < 8192)

// LDS accumulator of one component: sum I, sum (2x+1) I, sum (2y+1) I, peak = I << 32 | ~k
struct ChainAcc {
    unsigned long long sum_i, sum_xi, sum_yi, peak;
    uint32_t x_min, x_max, y_min, y_max;
    uint32_t num_pixels, pad;
};

// dynamic LDS: tile offsets | row offsets | forest (k, parent) later accumulators + record staging | word staging
constexpr int kChainForestBytes = kChainLdsEntries * 4;
constexpr int kChainAccBytes = kChainSlots * ((int)sizeof(ChainAcc) + (int)sizeof(WireRec2));
constexpr int kChainOutGlobalBytes = kChainThreads * (int)sizeof(WireRec2);
static_assert(kChainAccBytes <= kChainForestBytes && kChainOutGlobalBytes <= kChainForestBytes, "LDS plan");
constexpr int kChainStageOff = (kChainMaxTiles + 1) * 4 + (kChainMaxRows + 1) * 4 + kChainForestBytes;
constexpr int kChainSegOff = kChainStageOff + kChainWaves * 2 * kChainListCap * 4;
constexpr int kChainDynBytes = kChainSegOff + kChainWaves * kChainSegCap * 2;
static_assert(kChainStageOff % 8 == 0 && kChainDynBytes <= 160 * 1024 - 256, "LDS plan");
static_assert(2 * kChainRunCap * 4 <= kChainForestBytes + (kChainDynBytes - kChainStageOff), "LDS plan (runs)");
static_assert(kChainMaxRows < 8192, "run descriptor");

// (wave_inclusive_scan / wave_inclusive_max: kernels_threshold.hpp)

struct ChainArgs {
    CclArgs c;
    SegArgs s;
    uint32_t* h_counts;     // pinned host block (device address): [B] strong pixels, [B] components, [B][8] summary, [1] unused, [B] flags
    uint32_t max_batch;     // B
    uint32_t rec_stride;    // records between the frames' record areas (= max_comp)
    ThresholdArgs t;        // for the bright-window fix-up (fix_bright != 0): what k_bright_fix does, by this launch
    int fix_bright;
    uint32_t* fix_done;     // workgroups through with the bright list (the last one zeroes the list's count)
    int stop_after;         // (timing experiments) 1..4: return after phase A / E / U / P
    int runs_ok;            // 1: frames beyond kChainLdsEntries strong pixels take the run-based phases (16-bit pixels, W <= kChainRunMaxW); 2: every frame
    unsigned long long* phase_ts;   // (timing experiments) [frames][8] device timestamps at the phase boundaries, or null
};

__device__ __forceinline__ void chain_record(const SegArgs& sa, uint32_t W, uint32_t num_pixels, unsigned long long sum_i,
                                             unsigned long long sum_xi, unsigned long long sum_yi, uint32_t x_min, uint32_t x_max,
                                             uint32_t y_min, uint32_t y_max, uint32_t peak_k, uint32_t peak_i, uint32_t* s_sm, WireRec2& o) {
    // center_of_mass(): double sums of (c + 0.5) * I, quotient narrowed to float
    // (connected_components.hpp:81-100).  sum (2c+1) I is an exact integer; * 0.5 is exact.
    const double tot = (double)sum_i;
    const float com_x = (float)((double)sum_xi * 0.5 / tot), com_y = (float)((double)sum_yi * 0.5 / tot);
    const float com_z = (float)(0.5 * tot / tot);  // z = 0 for 2D (:247)
    const uint32_t peak_y = peak_k / W, peak_x = peak_k - peak_y * W;
    // peak_centroid_distance(): float arithmetic, connected_components.hpp:194-198; one rounding per
    // operation (no contraction in this library), float sqrt through the correctly rounded double sqrt
    const float dx = ((float)peak_x + 0.5f) - com_x;
    const float dy = ((float)peak_y + 0.5f) - com_y;
    const float dz = ((float)0 + 0.5f) - com_z;
    const float s2 = (dx * dx + dy * dy) + dz * dz;
    const float pcd = (float)__builtin_sqrt((double)s2);
    uint32_t flags = 0;
    // filter_reflections(): size first, then separation (connected_components.cc:207-236)
    if (sa.min_spot_size > 0 && num_pixels < sa.min_spot_size) flags |= 1u;
    else if (sa.max_sep > 0.0f && pcd > sa.max_sep) flags |= 2u;
    o.x_min = (uint16_t)x_min; o.x_max = (uint16_t)x_max; o.y_min = (uint16_t)y_min; o.y_max = (uint16_t)y_max;
    o.npx_flags = num_pixels | (flags << 30);
    o.com_x = com_x; o.com_y = com_y;
    o.peak_x = (uint16_t)peak_x; o.peak_y = (uint16_t)peak_y;
    o.peak_intensity = peak_i;
    o.peak_centroid_distance = pcd;
    o.sum_intensity = sum_i;
    // generate_boxes() filter (connected_components.cc:122-138)
    if (sa.min_spot_size == 0 || num_pixels >= sa.min_spot_size) {
        atomicAdd(&s_sm[0], 1u);
        atomicAdd(&s_sm[1], num_pixels);
    }
    if (flags == 0) atomicAdd(&s_sm[2], 1u);
    if (flags & 1u) atomicAdd(&s_sm[3], 1u);
    if (flags & 2u) atomicAdd(&s_sm[4], 1u);
}

// RUNS: the instantiation that also holds the run-based phases for frames beyond kChainLdsEntries strong pixels (launched when
// the stream's previous batch was that dense; a kernel of its own so that the usual one keeps its register allocation).
// LOG: the instantiation that takes the frame's strong pixels from the streaming kernel's wave logs (ThresholdArgs::wlog,
// tuning "strong_log") instead of the bit plane: phases B, A and E are replaced by L1 (undecided pixels decided, per-row
// counts) and, after S, L2 (the list placed in raster order by merging the logs of the strips of every band).
template <typename PixelT, bool RUNS = false, bool LOG = false>
__global__ __launch_bounds__(kChainThreads) void k_frame_chain(const ChainArgs A) {
    const CclArgs& a = A.c;
    const SegArgs& sa = A.s;
    extern __shared__ __align__(16) uint8_t s_dyn[];
    uint32_t* s_toff = reinterpret_cast<uint32_t*>(s_dyn);              // [n_tiles + 1]
    uint32_t* s_row = s_toff + (kChainMaxTiles + 1);                     // [H + 1] per-row counts, then offsets
    uint8_t* s_big = reinterpret_cast<uint8_t*>(s_row + (kChainMaxRows + 1));
    __shared__ uint32_t s_wave[kChainWaves];
    __shared__ uint32_t s_wrun[kChainWaves];   // runs listed by each wave (run-based phases)
    __shared__ uint32_t s_sm[8];
    __shared__ uint32_t s_flag;   // flags raised by any thread of the block (LOG)
    __shared__ uint32_t s_nmaybe;   // LOG: undecided pixels listed
    __shared__ uint32_t s_lst[kChainWaves][16];   // LOG: per chain wave, where each strip's entries start in the band's row of entries

    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FFS_PHASE_TS(A, 0);
    const int n_tiles = a.n_tiles;
    const uint32_t W = (uint32_t)a.W, H = (uint32_t)a.H;
    const int dpr = a.mpitch >> 2;

    // ---- B: bright windows (k_bright_fix's work): the listed pixels of THIS frame get their exact 64-bit sums and, if
    // strong, their plane bit, occupancy bit and tile count -- before phase A reads the counts
    uint32_t bright_flag = 0;
    if (!LOG && A.fix_bright) {
        const ThresholdArgs& t = A.t;
        const uint32_t listed = *t.bright_n;
        const uint32_t nb = min(listed, t.bright_cap);
        const uint8_t* fimg = (const uint8_t*)t.image + (uint64_t)frame * t.frame_stride;
        for (uint32_t e = tid; e < nb; e += kChainThreads) {
            const uint2 r = t.bright_list[e];
            if ((r.x >> 16) != (uint32_t)frame) continue;
            const uint32_t x = r.x & 0xFFFFu, y = r.y;
            if (exact_strong<PixelT, false>(t, fimg, (int)x, (int)y)) {
                uint8_t* plane = t.bits + (uint64_t)frame * t.plane_frame_stride + (uint64_t)y * t.mpitch;
                atomicOr(reinterpret_cast<uint32_t*>(plane) + (x >> 5), 1u << (x & 31u));
                atomicAdd(t.tile_counts + (uint64_t)frame * t.n_tiles + y / (uint32_t)kTileRows, 1u);
                const uint32_t ob = y * t.occ_spr + (x >> 7);
                atomicOr(t.occ + (uint64_t)frame * t.occ_frame_words + (ob >> 5), 1u << (ob & 31u));
            }
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            // the last workgroup through with the list empties it for the next batch (the host re-runs a batch whose list overflowed)
            const uint32_t before_me = atomicAdd(A.fix_done, 1u);
            if (before_me + 1 == gridDim.x) {
                if (listed > t.bright_cap) bright_flag = 8u;
                *t.bright_n = 0;
                *A.fix_done = 0;
            }
        }
    }

    // ---- A: tile offsets ------------------------------------------------------------------------------
    uint32_t total = 0;
    if constexpr (!LOG) {
        uint32_t* counts = sa.zero_counts + (uint64_t)frame * sa.zero_per_seg;   // (= a.tile_counts, writable)
        const uint32_t mine = tid < n_tiles ? counts[tid] : 0u;                  // (n_tiles <= kChainMaxTiles < kChainThreads)
        const uint32_t at = block_exclusive_scan<kChainThreads>(mine, s_wave, total);
        if (tid < n_tiles) {
            s_toff[tid] = at;
            counts[tid] = 0;   // consumed: the next batch's streaming kernel adds into zeros
        }
        if (tid == 0) {
            s_toff[n_tiles] = total;
            if (frame == 0 && sa.zero_word && !A.fix_bright) *sa.zero_word = 0;
        }
    }
    if (tid < 8) s_sm[tid] = 0;
    if (tid == 0) { s_flag = 0; s_nmaybe = 0; }
    if (tid < kChainWaves) s_wrun[tid] = 0;
    for (int y = tid; y <= a.H; y += kChainThreads) s_row[y] = 0;
    __syncthreads();
    uint32_t n = min(total, a.cap);                              // (LOG: known after phase S)
    bool in_lds = n <= (uint32_t)kChainLdsEntries;
    // denser: runs instead of pixels where the frame allows it (block-uniform)
    const bool runs = !LOG && RUNS && sizeof(PixelT) == 2 && A.runs_ok != 0 && (!in_lds || A.runs_ok == 2);
    uint32_t run_flag = 0;   // 16: more runs than the LDS plan holds; 32: a wave log (or the list of undecided pixels) overflowed; 64: a frame beyond
                             // the LDS forest met in the wave logs (the host runs the batch again another way)
    FFS_STOP_AFTER(A, 1);
    FFS_PHASE_TS(A, 1);

    uint32_t* gk = a.list_k + (uint64_t)frame * a.cap;
    uint32_t* gi = a.list_i + (uint64_t)frame * a.cap;
    uint32_t* gpar = a.parent + (uint64_t)frame * a.cap;
    CompAcc2* gacc = a.acc2 + (uint64_t)frame * a.cap;
    const uint8_t* img = (const uint8_t*)a.image + (uint64_t)frame * a.frame_stride;
    // the forest of a frame held in LDS.  (Never through a pointer that might also be global: a FLAT access may complete
    // out of order with the global ones, so every wait after it has to be for ALL outstanding memory operations, and the
    // loads this kernel keeps in flight would be waited for one by one.)
    uint32_t* spar = reinterpret_cast<uint32_t*>(s_big);
    auto pixel_at = [&](uint32_t kv) -> uint32_t {
        const uint32_t y = kv / W, x = kv - y * W;
        return (uint32_t)*reinterpret_cast<const PixelT*>(img + (uint64_t)y * a.pitch + (uint64_t)x * sizeof(PixelT));
    };

    // ---- E: compaction --------------------------------------------------------------------------------
    // Nothing in this phase waits for a load it has just issued: the plane words come two batches ahead, and the
    // pixel VALUES are not fetched here at all (phase P loads them, twenty independent loads per thread).  What bounds
    // it is one CU's share of the memory system: the frame's 2.3 MB plane at ~20 GB/s.
    // ---- L1 (LOG): the frame's entries in the wave logs -- undecided pixels decided, per-row counts ------------------------
    // Frame `frame` = frame fe of super row y; its groups are G0 .. G0 + gpf - 1 of the super row, held by strips ls0 .. ls1
    // (kSOwned groups each); the log of band b of strip k of super row y lies at log_slot(y, b, k) (ffs_device.h).  A log is
    // sorted by (row, frame, group), the strips of a band partition the groups in order: the raster order of a row is strip
    // after strip.  Chain wave w takes bands w, w + 16, ...
    [[maybe_unused]] const ThresholdArgs& T = A.t;
    [[maybe_unused]] uint32_t l_y = 0, l_fe = 0, l_s0 = 0, l_ns = 0;
    [[maybe_unused]] auto log_wave = [&](int band, uint32_t k) -> uint32_t {
        return log_slot(T, l_y, (uint32_t)band, l_s0 + k);
    };
    // The logs of one band, read side by side: lane k fetches strip k's count (one round trip for all strips), a scan lays
    // the strips' entries end to end, and lane i of a chunk takes entry i of that row of entries -- whichever log it is in.
    // band_begin() returns the band's entries (all frames of the strips'); band_entry() maps a flat index to (strip, entry).
    [[maybe_unused]] auto band_begin = [&](int band) -> uint32_t {
        uint32_t c = 0;
        if ((uint32_t)lane < l_ns) {
            c = T.wlog_n[log_wave(band, (uint32_t)lane)];
            if (c > (uint32_t)kWlogCap) { atomicOr(&s_flag, 32u); c = (uint32_t)kWlogCap; }   // (a wave wanted more than its log holds)
        }
        const uint32_t incl = wave_inclusive_scan(c);
        __builtin_amdgcn_wave_barrier();
        if (lane < 16) s_lst[wave][lane] = incl - c;   // where strip `lane`'s entries start (strips beyond the last: the total)
        __builtin_amdgcn_wave_barrier();
        return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    };
    [[maybe_unused]] auto band_entry = [&](uint32_t f, uint32_t nb, uint32_t& k, uint32_t& e) -> bool {
        k = 0;
        for (uint32_t j = 1; j < l_ns; ++j) k += f >= s_lst[wave][j] ? 1u : 0u;   // (wave-uniform loop, LDS broadcast reads)
        e = f - s_lst[wave][k];
        return f < nb;
    };
    constexpr int kLogChunks = 8;
    [[maybe_unused]] constexpr uint32_t kGroupPx = sizeof(PixelT) == 2 ? 8u : 4u;   // pixels of a lane group of the streaming kernels
    // eight chunks (64 entries each) of the band's row of entries from `sb` on, all loads issued before any is used;
    // entries past the end read as "frame 0xFFFF" (nobody's)
    [[maybe_unused]] auto load_chunks = [&](int band, uint32_t sb, uint32_t nb, uint2 (&ent)[kLogChunks]) {
#pragma unroll
        for (int c = 0; c < kLogChunks; ++c) {
            uint32_t k, e;
            ent[c] = make_uint2(0xFFFFFFFFu, 0xFFFF0000u);
            if (band_entry(sb + (uint32_t)(c * 64 + lane), nb, k, e)) {
                // (agent-scope load: past this CU's L1, which the atomics that decide undecided pixels never refresh -- no fence needed,
                // and a device-wide fence here writes back and invalidates the L2 under the streaming kernel: -4 % on the step)
                const unsigned long long q = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(T.wlog + (uint64_t)log_wave(band, k) * kWlogCap + e),
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ent[c] = make_uint2((uint32_t)q, (uint32_t)(q >> 32));
            }
        }
    };
    if constexpr (LOG) {
        l_y = (uint32_t)frame / (uint32_t)T.group_frames;
        l_fe = (uint32_t)frame - l_y * (uint32_t)T.group_frames;
        const uint32_t gsep = (uint32_t)T.gpf + 1u, G0 = l_fe * gsep, G1 = G0 + (uint32_t)T.gpf - 1u;
        l_s0 = G0 / (uint32_t)kSOwned;
        l_ns = G1 / (uint32_t)kSOwned - l_s0 + 1u;
        // Undecided pixels (windows with sum p >= 65536: the cores of bright spots) are listed here and decided below, one per
        // thread -- decided where they are met, a lane with eight of them kept its whole wave waiting (118 us for this phase).
        uint32_t* s_maybe = reinterpret_cast<uint32_t*>(s_dyn + kChainStageOff);   // band << 16 | strip << 12 | entry << 3 | bit
        constexpr uint32_t kMaybeCap = (uint32_t)(kChainWaves * 2 * kChainListCap);   // the staging area: 10 240 words
        for (int band = wave; band < T.n_bands; band += kChainWaves) {
            const uint32_t nb = band_begin(band);
            for (uint32_t sb = 0; sb < nb; sb += 64 * kLogChunks) {   // wave-uniform; eight chunks of loads in flight (usually all there is)
                uint2 ent[kLogChunks];
                load_chunks(band, sb, nb, ent);
#pragma unroll
                for (int c = 0; c < kLogChunks; ++c) {
                    const uint2 v = ent[c];
                    if ((v.y >> 16) != l_fe) continue;   // (entries beyond the band's, and other frames': frame 0xFFFF)
                    const uint32_t row = v.x >> 16;
                    uint32_t maybe = (v.y >> 8) & 0xFFu & ~v.y;
                    if (maybe) {
                        uint32_t k, e;
                        (void)band_entry(sb + (uint32_t)(c * 64 + lane), nb, k, e);
                        const uint32_t at = atomicAdd(&s_nmaybe, (uint32_t)__popc(maybe));
                        uint32_t q = at;
                        while (maybe) {
                            const int b = __ffs((int)maybe) - 1;
                            maybe &= maybe - 1;
                            if (q < kMaybeCap) s_maybe[q] = ((uint32_t)band << 16) | (k << 12) | (e << 3) | (uint32_t)b;
                            ++q;
                        }
                    }
                    const uint32_t pc = (uint32_t)__popc(v.y & 0xFFu);
                    if (pc) atomicAdd(&s_row[row], pc);
                }
            }
        }
        __syncthreads();
        {
            const uint32_t nm = s_nmaybe;
            if (nm > kMaybeCap && tid == 0) atomicOr(&s_flag, 32u);
            for (uint32_t i = tid; i < min(nm, kMaybeCap); i += kChainThreads) {
                const uint32_t it = s_maybe[i];
                uint2* ent = T.wlog + (uint64_t)log_wave((int)(it >> 16), (it >> 12) & 15u) * kWlogCap + ((it >> 3) & 0x1FFu);
                const uint2 v = *ent;
                const uint32_t row = v.x >> 16, ge = v.x & 0xFFFFu, b = it & 7u;
                if (exact_strong<PixelT, false>(T, img, (int)(ge * kGroupPx + b), (int)row)) {
                    atomicOr(&ent->y, 1u << b);   // (phase L2 reads the entry again; the "undecided" bits stay and are ignored)
                    atomicAdd(&s_row[row], 1u);
                }
            }
        }
        __syncthreads();   // (also waits for this block's atomics to have been performed)
    }
    if (!LOG && total != 0) {
        const int tpw = (n_tiles + kChainWaves - 1) / kChainWaves;
        const int tb = min(wave * tpw, n_tiles), te = min(tb + tpw, n_tiles);
        const int yb = tb * kTileRows, ye = min(te * kTileRows, a.H);
        const int nw = (ye - yb) * dpr;                       // plane words of this wave's rows (contiguous)
        if (te > tb && s_toff[te] != s_toff[tb]) {            // wave-uniform
            uint32_t* words = reinterpret_cast<uint32_t*>(a.bits + (uint64_t)frame * a.plane_frame_stride + (uint64_t)yb * a.mpitch);
            uint8_t* sbytes = a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride;
            uint32_t* s_g = reinterpret_cast<uint32_t*>(s_dyn + kChainStageOff) + wave * 2 * kChainListCap;
            uint32_t* s_w = s_g + kChainListCap;

            uint32_t run = s_toff[tb];             // list position of the next strong pixel (wave-uniform)
            // run-based phases: this wave's run descriptors go, in order, to entries s_toff[tb] .. of the array the pixel path
            // keeps its parents in (there are never more runs than pixels, so the waves' ranges cannot meet)
            uint32_t rrun = s_toff[tb];            // scratch position of the next run (wave-uniform)
            int n_list = 0;                        // staged non-zero words (wave-uniform)
            uint32_t last_g = 0xFFFFFFFFu, last_w = 0;  // the word before the staged ones (for the link to the left)
            // places the pixels of `cnt` (<= 64) staged words from `first` on, one word per lane
            auto place = [&](int first, int cnt) {
                const bool valid = lane < cnt;
                const int e = first + lane;
                const uint32_t g = valid ? s_g[e] : 0u;
                uint32_t w = valid ? s_w[e] : 0u;
                const uint32_t pg = e > 0 ? s_g[valid ? e - 1 : 0] : last_g, pw = e > 0 ? s_w[valid ? e - 1 : 0] : last_w;
                const uint32_t pc = (uint32_t)__popc(w);
                const uint32_t inc = wave_inclusive_scan(pc);
                uint32_t at = run + inc - pc;
                run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                if constexpr (RUNS) if (runs) {   // block-uniform
                    // runs inside the word (one that continues from the word before starts anew: phase U' joins them)
                    const uint32_t starts = w & ~(w << 1);
                    const uint32_t rc = (uint32_t)__popc(starts);
                    const uint32_t rinc = wave_inclusive_scan(rc);
                    uint32_t rat = rrun + rinc - rc;
                    rrun += (uint32_t)__builtin_amdgcn_readlane((int)rinc, 63);
                    if (valid) {
                        const int row = (int)g / dpr;
                        const int col = (int)g - row * dpr;
                        const int xb = col * 32;
                        const int y = yb + row;
                        atomicAdd(&s_row[y], rc);   // runs per row here
                        uint32_t st = starts;
                        while (st) {
                            const int b = __ffs((int)st) - 1;
                            st &= st - 1;
                            const uint32_t rest = w >> b;   // bit 0 = the run's first pixel
                            const uint32_t len = rest == 0xFFFFFFFFu ? 32u : (uint32_t)__ffs((int)~rest) - 1u;
                            if (rat < a.cap) gpar[rat] = ((uint32_t)y << 19) | ((uint32_t)(xb + b) << 5) | (len - 1u);
                            ++rat;
                        }
                        while (w) {
                            const int b = __ffs((int)w) - 1;
                            w &= w - 1;
                            if (a.need_lists && at < a.cap) gk[at] = (uint32_t)y * W + (uint32_t)(xb + b);   // (the run-based phases never read the pixel list)
                            if (a.dense_bytes) sbytes[(uint64_t)y * a.bpitch + (uint32_t)(xb + b)] = 1;
                            ++at;
                        }
                        if (a.clear_bits) words[g] = 0;
                    }
                    return;
                }
                if (valid) {
                    if (a.clear_bits) words[g] = 0;   // (the streaming kernel needs an all-zero plane)
                    const int row = (int)g / dpr;
                    const int col = (int)g - row * dpr;
                    const int xb = col * 32;
                    const int y = yb + row;
                    atomicAdd(&s_row[y], pc);
                    // the pixel left of this word's bit 0: bit 31 of the word before it, same row
                    const bool left = col != 0 && pg + 1 == g && (pw >> 31) != 0u;
                    int prev = -2;  // bit index of this word's previous strong pixel
                    uint32_t run_first = at;  // list index of the first pixel of the current run inside this word
                    while (w) {
                        const int b = __ffs((int)w) - 1;
                        w &= w - 1;
                        const bool linked = b == 0 ? left : prev == b - 1;
                        if (!linked || b == 0) run_first = at;
                        if (at < a.cap) {
                            // a run's pixels point at its first pixel in this word; a run that continues from
                            // the previous word hooks on to that word's last pixel (see emit_tile_w)
                            const uint32_t pv = !linked ? at : (b == 0 ? at - 1 : run_first);
                            gk[at] = (uint32_t)y * W + (uint32_t)(xb + b);
                            if (in_lds) {
                                if (at < (uint32_t)kChainLdsEntries) spar[at] = pv;
                            } else {
                                gpar[at] = pv;
                                if (!linked) {  // a run start may end up a root: fresh accumulator
                                    CompAcc2 z;
                                    z.sum_i = z.sum_xi = z.sum_yi = z.peak = 0ull;
                                    z.x_min = 0xFFFFFFFFu; z.x_max = 0u; z.y_min = 0xFFFFFFFFu; z.y_max = 0u;
                                    z.num_pixels = 0u; z.pad = 0u;
                                    gacc[at] = z;
                                }
                            }
                        }
                        if (a.dense_bytes) sbytes[(uint64_t)y * a.bpitch + (uint32_t)(xb + b)] = 1;  // the reference kernel's result_strong byte
                        ++at;
                        prev = b;
                    }
                }
            };
            // places every full group of 64 staged words (all of them at the end of the range); what is left (< 64)
            // moves to the front
            auto drain = [&](bool all) {
                int done = 0;
                while (n_list - done >= (all ? 1 : 64)) {
                    const int cnt = min(64, n_list - done);
                    place(done, cnt);
                    done += cnt;
                }
                if (done == 0) return;
                last_g = s_g[done - 1];
                last_w = s_w[done - 1];
                const int rem = n_list - done;
                const uint32_t mg = lane < rem ? s_g[done + lane] : 0u, mw = lane < rem ? s_w[done + lane] : 0u;
                if (lane < rem) {
                    s_g[lane] = mg;
                    s_w[lane] = mw;
                }
                n_list = rem;
            };

            const rsrc_t r_words = make_rsrc(words, (uint32_t)nw * 4u);   // (out of range reads 0: no branch around a load)
            if (a.use_occ) {
                // The streaming kernel left one bit per 16-byte segment of the plane that holds a strong pixel: read the
                // bitmap (a few KB per wave instead of its 140 KB of plane), list the occupied segments in order, load
                // only those.  The bits are cleared as they are read, like the plane words.
                uint32_t* occ = a.occ + (uint64_t)frame * a.occ_frame_words;
                uint16_t* s_seg = reinterpret_cast<uint16_t*>(s_dyn + kChainSegOff) + wave * kChainSegCap;
                const uint32_t spr = a.occ_spr;
                const uint32_t b0 = (uint32_t)yb * spr, nbits = (uint32_t)(ye - yb) * spr;
                auto fetch8 = [&](uint32_t base, uint32_t& lo, uint32_t& hi) {   // the two bitmap words this lane's 8 bits lie in
                    const uint32_t rel = base + 8u * (uint32_t)lane;
                    lo = hi = 0;
                    if (rel < nbits) {
                        const uint32_t wi = (b0 + rel) >> 5;
                        lo = occ[wi];
                        hi = occ[wi + 1];   // (the bitmap has two words of slack)
                    }
                };
                uint32_t lo, hi;
                fetch8(0, lo, hi);
                for (uint32_t base = 0; base < nbits; base += 512u) {
                    const uint32_t rel = base + 8u * (uint32_t)lane;
                    uint32_t v = 0;
                    if (rel < nbits) {
                        const uint32_t sh = (b0 + rel) & 31u, nb = min(8u, nbits - rel);
                        const uint32_t mask = (1u << nb) - 1u;
                        v = (uint32_t)((((unsigned long long)hi << 32) | lo) >> sh) & mask;
                        if (v) {   // consumed: clear (neighbouring lanes, waves and 8-bit groups share the words)
                            const uint32_t wi = (b0 + rel) >> 5;
                            atomicAnd(occ + wi, ~(v << sh));
                            if (sh > 24u) atomicAnd(occ + wi + 1, ~(v >> (32u - sh)));
                        }
                    }
                    uint32_t nlo, nhi;
                    fetch8(base + 512u, nlo, nhi);   // the next round's words are on their way while this one is worked on
                    const uint32_t cnt = (uint32_t)__popc(v);
                    const uint32_t inc = wave_inclusive_scan(cnt);
                    const int T = __builtin_amdgcn_readlane((int)inc, 63);
                    if (T != 0) {
                        uint32_t pos = inc - cnt;
                        while (v) {
                            const int b = __ffs((int)v) - 1;
                            v &= v - 1;
                            s_seg[pos++] = (uint16_t)(rel + (uint32_t)b);
                        }
                        for (int sb = 0; sb < T; sb += 64) {
                            const int e = sb + lane;
                            const bool valid = e < T;
                            const uint32_t sbit = valid ? (uint32_t)s_seg[e] : 0u;
                            const uint32_t row_rel = sbit / spr, seg = sbit - row_rel * spr;
                            const uint32_t woff = row_rel * (uint32_t)dpr + seg * 4u;   // word offset in this wave's rows
                            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r_words, valid ? woff * 4u : 0x80000000u, 0, 0);
                            const uint32_t vv[4] = {q[0], q[1], q[2], q[3]};
                            const uint32_t cw = (vv[0] != 0u) + (vv[1] != 0u) + (vv[2] != 0u) + (vv[3] != 0u);
                            if (n_list + 256 > kChainListCap) drain(false);
                            const uint32_t iw = wave_inclusive_scan(cw);
                            int at = n_list + (int)(iw - cw);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (vv[c]) {
                                    s_g[at] = woff + (uint32_t)c;
                                    s_w[at] = vv[c];
                                    ++at;
                                }
                            }
                            n_list += __builtin_amdgcn_readlane((int)iw, 63);
                        }
                    }
                    lo = nlo;
                    hi = nhi;
                }
                drain(true);
            } else {
            const int nrounds = (nw + 255) / 256;   // a round = 4 x 64 consecutive words: lane L holds words 256 r + 64 c + L
            // Rounds whose words lie in tiles without a strong pixel are not loaded at all (wave-uniform test).  Rounds are
            // visited in increasing order, so the tile of a round's first word is tracked, not divided for.
            const int wpt = kTileRows * dpr;        // plane words per tile
            int lt = 0;                             // tile (relative to tb) of the first word of the next round to load
            auto round_live = [&](int r) {
                if (r >= nrounds) return false;
                const int w0 = r * 256, w1 = min(w0 + 255, nw - 1);
                while ((lt + 1) * wpt <= w0) ++lt;
                int t1 = lt;
                while ((t1 + 1) * wpt <= w1) ++t1;
                return s_toff[tb + t1 + 1] != s_toff[tb + lt];
            };
            // (buffer loads: out of range -- beyond the wave's words, or switched off -- reads 0, so the loads are not
            // wrapped in branches and the waits below can count them)
            auto load_batch = [&](int b, uint32_t (&buf)[kChainQuads][4]) {
#pragma unroll
                for (int q = 0; q < kChainQuads; ++q) {
                    const int r = b * kChainQuads + q;
                    const uint32_t off = round_live(r) ? (uint32_t)(r * 256 + lane) * 4u : 0x80000000u;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        buf[q][c] = __builtin_amdgcn_raw_buffer_load_b32(r_words, off, (uint32_t)c * 256u, 0);
                }
            };
            // (no scan here: the 64 words of a quarter round are consecutive, so ballot + mbcnt keeps the order;
            // and no global store, so the loads in flight are waited for by count, not all together)
            auto stage_batch = [&](int b, const uint32_t (&buf)[kChainQuads][4]) {
#pragma unroll
                for (int q = 0; q < kChainQuads; ++q) {
                    const int r = b * kChainQuads + q;
                    if (__builtin_amdgcn_ballot_w64((buf[q][0] | buf[q][1] | buf[q][2] | buf[q][3]) != 0u) == 0ull) continue;  // wave-uniform
                    if (n_list + 256 > kChainListCap) drain(false);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t w = buf[q][c];
                        const unsigned long long nz = __builtin_amdgcn_ballot_w64(w != 0u);
                        if (nz == 0ull) continue;  // wave-uniform
                        if (w) {
                            const int e = n_list + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));
                            s_g[e] = (uint32_t)(r * 256 + c * 64 + lane);
                            s_w[e] = w;
                        }
                        n_list += __popcll(nz);
                    }
                }
            };
            // three batches of loads live: the one being staged and two in flight
            uint32_t b0[kChainQuads][4], b1[kChainQuads][4], b2[kChainQuads][4];
            const int nbatches = (nrounds + kChainQuads - 1) / kChainQuads;
            load_batch(0, b0);
            load_batch(1, b1);
            for (int b = 0; b < nbatches; b += 3) {   // (beyond the last batch nothing is live: no load is issued)
                load_batch(b + 2, b2);
                stage_batch(b, b0);
                load_batch(b + 3, b0);
                stage_batch(b + 1, b1);
                load_batch(b + 4, b1);
                stage_batch(b + 2, b2);
            }
            drain(true);
            }
            if (lane == 0) s_wrun[wave] = rrun - s_toff[tb];
        }
    }
    __syncthreads();
    FFS_STOP_AFTER(A, 2);
    FFS_PHASE_TS(A, 2);

    // ---- S: per-row counts -> list offset of the first strong pixel of every row (and of "row H" = n) ------
    {
        constexpr int kPerRow = (kChainMaxRows + 1 + kChainThreads - 1) / kChainThreads;   // consecutive rows per thread
        const int y0 = min(tid * kPerRow, a.H + 1), y1 = min(y0 + kPerRow, a.H + 1);
        uint32_t loc[kPerRow];
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < kPerRow; ++q) {
            loc[q] = y0 + q < y1 ? s_row[y0 + q] : 0u;
            mine += loc[q];
        }
        uint32_t tot;
        uint32_t run = block_exclusive_scan<kChainThreads>(mine, s_wave, tot);
#pragma unroll
        for (int q = 0; q < kPerRow; ++q) {
            if (y0 + q < y1) {
                s_row[y0 + q] = min(run, a.cap);
                run += loc[q];
            }
        }
        if constexpr (LOG) {
            total = tot;
            n = min(total, a.cap);
            in_lds = n <= (uint32_t)kChainLdsEntries;
        }
    }
    __syncthreads();
    if constexpr (LOG) {
        // ---- L2: the list in raster order.  Per band (one chain wave each): pixels per (row, strip) -> where each strip's part
        // of each row starts -> every entry's pixels placed (a segmented scan over the 64 entries of a chunk of a log).  The
        // column numbers go straight into LDS (phase X has nothing left to do).  Frames beyond the LDS forest: flag 32.
        run_flag |= s_flag;
        if (!in_lds) run_flag |= 64u;   // (a frame beyond the LDS forest: this batch again through the plane; logs again when the data is sparse again)
        if (run_flag == 0 && total != 0) {
            constexpr int kMaxStrips = 16;
            uint16_t* s_x = reinterpret_cast<uint16_t*>(s_dyn + kChainStageOff);
            // per chain wave: cw[row of the band][strip] in the forest area (free until the forest is built)
            uint32_t* cw = reinterpret_cast<uint32_t*>(s_big) + (size_t)wave * (kChainForestBytes / 4 / kChainWaves);
            const int cw_rows = (kChainForestBytes / 4 / kChainWaves) / kMaxStrips;   // 106 rows of a band at a time
            uint8_t* sbytes = a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride;
            const uint32_t ns = min(l_ns, (uint32_t)kMaxStrips);
            for (int band = wave; band < T.n_bands; band += kChainWaves) {
                const int yb0 = band_first_row(band, T.band_rows, T.band_rows2, T.band_split);
                const int yb1 = min(band_first_row(band + 1, T.band_rows, T.band_rows2, T.band_split), (int)a.H);
                for (int r0 = yb0; r0 < yb1; r0 += cw_rows) {   // (bands taller than the counters: in pieces)
                    const int r1 = min(r0 + cw_rows, yb1);
                    for (int i = lane; i < (r1 - r0) * kMaxStrips; i += 64) cw[i] = 0;
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t nb = band_begin(band);
                    const bool one = nb <= 64u * kLogChunks;   // (the usual case: the entries stay in registers between the two passes)
                    uint2 ent[kLogChunks];
                    for (uint32_t sb = 0; sb < nb; sb += 64 * kLogChunks) {   // pixels per (row, strip)
                        load_chunks(band, sb, nb, ent);
#pragma unroll
                        for (int c = 0; c < kLogChunks; ++c) {
                            const uint2 v = ent[c];
                            const int row = (int)(v.x >> 16);
                            if ((v.y >> 16) != l_fe || row < r0 || row >= r1) continue;
                            uint32_t k, e;
                            (void)band_entry(sb + (uint32_t)(c * 64 + lane), nb, k, e);
                            if (k < ns) atomicAdd(&cw[(row - r0) * kMaxStrips + (int)k], (uint32_t)__popc(v.y & 0xFFu));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (int r = r0 + lane; r < r1; r += 64) {   // counts -> list positions (the row's own offset + the strips before)
                        uint32_t at = s_row[r];
                        for (uint32_t k = 0; k < ns; ++k) {
                            const uint32_t c = cw[(r - r0) * kMaxStrips + (int)k];
                            cw[(r - r0) * kMaxStrips + (int)k] = at;
                            at += c;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t sb = 0; sb < nb; sb += 64 * kLogChunks) {
                        if (!one) load_chunks(band, sb, nb, ent);
#pragma unroll
                        for (int c = 0; c < kLogChunks; ++c) {   // 64 consecutive entries of the band's row of entries
                            if (sb + (uint32_t)(c * 64) >= nb) break;   // wave-uniform
                            const uint2 v = ent[c];
                            uint32_t k, e;
                            const bool have = band_entry(sb + (uint32_t)(c * 64 + lane), nb, k, e);
                            const int row = (int)(v.x >> 16);
                            const bool mine = have && (v.y >> 16) == l_fe && row >= r0 && row < r1 && k < ns;
                            // the group's pixels, as the streaming kernel had them (on their way while the scans below run): the
                            // intensity list is written here, in list order, and phase P never gathers from the image
                            uint4 px = make_uint4(0u, 0u, 0u, 0u);
                            if (mine) px = T.wpix[(uint64_t)log_wave(band, k) * kWlogCap + e];
                            const uint32_t cb = mine ? v.y & 0xFFu : 0u;
                            const uint32_t pc = (uint32_t)__popc(cb);
                            const uint32_t incl = wave_inclusive_scan(pc), excl = incl - pc;
                            // the pixels of MY (strip, row) before me in this chunk: excl minus excl at the first lane of my segment (a
                            // log is sorted by row and the logs follow each other: a segment is contiguous; lanes that are not mine
                            // carry keys of their own and no pixels)
                            const uint32_t key = mine ? (k << 16) | (uint32_t)row : 0x80000000u + (uint32_t)lane;
                            const uint32_t key_prev = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)key, 0x138, 0xf, 0xf, false);   // wave_shr:1
                            const bool starts = lane == 0 || key_prev != key;
                            const uint32_t seg = wave_inclusive_max(starts ? excl : 0u);
                            if (mine) {
                                uint32_t at = cw[(row - r0) * kMaxStrips + (int)k] + (excl - seg);
                                const uint32_t ge = v.x & 0xFFFFu;
                                uint32_t w = cb;
                                while (w) {
                                    const int b = __ffs((int)w) - 1;
                                    w &= w - 1;
                                    const uint32_t x = ge * kGroupPx + (uint32_t)b;
                                    uint32_t I;
                                    if constexpr (sizeof(PixelT) == 2) {
                                        const uint32_t pw = b < 2 ? px.x : b < 4 ? px.y : b < 6 ? px.z : px.w;
                                        I = (b & 1) ? pw >> 16 : pw & 0xFFFFu;
                                    } else {
                                        I = b == 0 ? px.x : b == 1 ? px.y : b == 2 ? px.z : px.w;
                                    }
                                    if (at < a.cap) gi[at] = I;
                                    if (a.need_lists && at < a.cap) gk[at] = (uint32_t)row * W + x;
                                    if (at < (uint32_t)kChainLdsEntries) s_x[at] = (uint16_t)x;
                                    if (a.dense_bytes) sbytes[(uint64_t)row * a.bpitch + x] = 1;
                                    ++at;
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            // the next chunk may continue the chunk's last segment: move that (row, strip)'s position on
                            const uint32_t key_next = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)key, 0x130, 0xf, 0xf, false);   // wave_shl:1
                            if (mine && (lane == 63 || key_next != key)) cw[(row - r0) * kMaxStrips + (int)k] += incl - seg;
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }
        }
        if (run_flag) n = 0;   // (nothing was placed: the phases below have nothing to do)
        __syncthreads();
        FFS_STOP_AFTER(A, 5);
        FFS_PHASE_TS(A, 3);
    }

    WireRec2* recs = reinterpret_cast<WireRec2*>(sa.recs) + (uint64_t)frame * A.rec_stride;
    uint32_t before = 0;  // components numbered so far (block-uniform)

    if (in_lds && !runs) {
        // every thread owns `per` consecutive list entries (<= kChainPer) through phases U, P and R
        const uint32_t per = (n + kChainThreads - 1) / kChainThreads;
        const uint32_t i0 = min((uint32_t)tid * per, n), i1 = min(i0 + per, n);

        // ---- X: the list's column numbers into LDS (16 bits each, in the staging area phase E is done with) --------
        // With the per-row list offsets they ARE the list (k = y W + x), so phases U and P make no global access to it:
        // a dependent access costs ~0.1 us here against ~1 us at the L2.
        uint16_t* s_x = reinterpret_cast<uint16_t*>(s_dyn + kChainStageOff);
        if constexpr (!LOG) {   // (phase L2 wrote them as it placed the list)
            for (uint32_t i = tid; i < n; i += kChainThreads) {
                const uint32_t kv = gk[i];
                s_x[i] = (uint16_t)(kv - (kv / W) * W);
            }
            __syncthreads();
        }
        // the row of list entry i0: the last row whose offset is <= i0 (rows without strong pixels share their successor's)
        uint32_t yrow = 0;
        if (i0 < i1) {
            uint32_t lo = 0, hi = H;   // first r in (0, H] with s_row[r] > i0, minus one
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (s_row[mid + 1] <= i0) lo = mid + 1; else hi = mid;
            }
            yrow = lo;
        }
        // does entry j (in row r) continue the run of entry j - 1, i.e. k[j - 1] + 1 == k[j]?  (Across a row end too:
        // the reference's k + 1 edge has no row-end check, connected_components.cc:62-70.)
        auto continues = [&](uint32_t j, uint32_t r) -> bool {
            if (j == 0) return false;
            const uint32_t xj = s_x[j], xp = s_x[j - 1];
            if (j != s_row[r]) return xp + 1 == xj;                                   // same row
            return xj == 0 && xp == W - 1 && r > 0 && s_row[r - 1] < s_row[r];        // first of its row: the row above ends the list before it
        };

        if constexpr (LOG) {
            // the forest's run links, which the plane's compaction sets as it places a word's pixels: an entry whose left
            // neighbour is strong points at it (the row-wrap pair is joined in phase U, as there)
            uint32_t y = yrow;
            for (uint32_t i = i0; i < i1; ++i) {
                while (s_row[y + 1] <= i) ++y;
                spar[i] = (continues(i, y) && s_x[i] != 0) ? i - 1 : i;
            }
            __syncthreads();
        }
        // ---- U: vertical edges + row wrap (k_union<false> with the runs linked by the compaction) ----------------
        {
            uint32_t y = yrow;
            for (uint32_t i = i0; i < i1; ++i) {
                while (s_row[y + 1] <= i) ++y;
                const uint32_t x = s_x[i];
                const bool cont = continues(i, y);
                if (cont && x == 0) uf_union(spar, i - 1, i);   // the row-wrap edge (a run inside a row was linked by the compaction)
                if (y + 1 >= H) continue;
                uint32_t lo = max(i + 1, s_row[y + 1]);
                const uint32_t end = max(lo, min(min(n, i + 1 + W), s_row[y + 2]));
                uint32_t hi = end;
                while (lo < hi) {   // lower bound of x among the next row's entries
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    if ((uint32_t)s_x[mid] < x) lo = mid + 1; else hi = mid;
                }
                if (lo < end && (uint32_t)s_x[lo] == x) {
                    // one edge per pair of overlapping runs is enough (see k_union)
                    if (!cont || !continues(lo, y + 1)) uf_union(spar, i, lo);
                }
            }
        }
        __syncthreads();
        FFS_STOP_AFTER(A, 3);
        FFS_PHASE_TS(A, 4);

        // ---- P: pixel values; roots, numbered in list order ---------------------------------------------------------
        // This thread's entries stay in registers through phase R: k, intensity and a 16-bit id (first the root's list
        // index, then the component's number; both < kChainLdsEntries < 0xFFFF = none).  16-bit pixels share a register
        // with the id, 32-bit pixels keep their ids two to a register.
        constexpr bool kPack = sizeof(PixelT) == 2;
        uint32_t exy[kChainPer], ew[kChainPer], eid[kPack ? 1 : kChainPer / 2];   // (y << 16 | x: H <= kChainMaxRows, W <= 65535)
        auto get_id = [&](int q) -> uint32_t {
            if constexpr (kPack) return ew[q] >> 16;
            else return (eid[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
        };
        auto set_id = [&](int q, uint32_t v) {
            if constexpr (kPack) ew[q] = (ew[q] & 0xFFFFu) | (v << 16);
            else eid[q >> 1] = (eid[q >> 1] & ~(0xFFFFu << (16 * (q & 1)))) | (v << (16 * (q & 1)));
        };
        auto get_i = [&](int q) -> uint32_t { return kPack ? ew[q] & 0xFFFFu : ew[q]; };
        if constexpr (!kPack) {
#pragma unroll
            for (int q = 0; q < kChainPer / 2; ++q) eid[q] = 0xFFFFFFFFu;
        }
        constexpr uint32_t kOob = 0x80000000u;   // (an offset no resource reaches: the load returns 0)
        {
            uint32_t y = yrow;
#pragma unroll
            for (int q = 0; q < kChainPer; ++q) {
                exy[q] = 0;
                if (i0 + q < i1) {
                    while (s_row[y + 1] <= i0 + q) ++y;
                    exy[q] = (y << 16) | (uint32_t)s_x[i0 + q];
                }
            }
        }
        const rsrc_t r_img = make_rsrc(img, (uint32_t)a.H * a.pitch);
        if constexpr (LOG) {
            // (phase L2 wrote the intensity list from the logs: this thread's entries are consecutive dwords of it)
            const rsrc_t r_gi = make_rsrc(gi, a.cap * 4u);
#pragma unroll
            for (int q = 0; q < kChainPer; ++q)
                ew[q] = kPack ? (__builtin_amdgcn_raw_buffer_load_b32(r_gi, i0 + q < i1 ? (i0 + (uint32_t)q) * 4u : kOob, 0, 0) & 0xFFFFu) | 0xFFFF0000u
                              : __builtin_amdgcn_raw_buffer_load_b32(r_gi, i0 + q < i1 ? (i0 + (uint32_t)q) * 4u : kOob, 0, 0);
        } else {
#pragma unroll
        for (int q = 0; q < kChainPer; ++q) {
            const uint32_t y = exy[q] >> 16, x = exy[q] & 0xFFFFu;
            const uint32_t off = i0 + q < i1 ? y * a.pitch + x * (uint32_t)sizeof(PixelT) : kOob;
            uint32_t v;
            if constexpr (kPack) v = ((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r_img, off, 0, 0) & 0xFFFFu) | 0xFFFF0000u;
            else v = __builtin_amdgcn_raw_buffer_load_b32(r_img, off, 0, 0);
            ew[q] = v;
        }
        }
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < kChainPer; ++q) {
            if (i0 + q < i1) {
                const uint32_t root = uf_find(spar, i0 + q);
                set_id(q, root);
                mine += root == i0 + q ? 1u : 0u;
                if constexpr (!LOG) gi[i0 + q] = get_i(q);
            }
        }
        __syncthreads();   // every find is done: the forest's root slots now take the component numbers
        {
            uint32_t slot = block_exclusive_scan<kChainThreads>(mine, s_wave, before);
#pragma unroll
            for (int q = 0; q < kChainPer; ++q)
                if (i0 + q < i1 && get_id(q) == i0 + q) spar[i0 + q] = slot++;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kChainPer; ++q)
            if (i0 + q < i1) set_id(q, spar[get_id(q)]);
        __syncthreads();   // the forest is dead from here on: its LDS becomes accumulators + record staging
        FFS_STOP_AFTER(A, 4);
        FFS_PHASE_TS(A, 5);

        // ---- R: kChainSlots components at a time --------------------------------------------------------------
        ChainAcc* s_acc = reinterpret_cast<ChainAcc*>(s_big);
        uint32_t* s_out = reinterpret_cast<uint32_t*>(s_big + kChainSlots * sizeof(ChainAcc));
        const uint32_t ncomp = min(before, sa.max_comp);
        for (uint32_t c0 = 0; c0 < ncomp; c0 += kChainSlots) {
            if (tid < kChainSlots) {
                ChainAcc z;
                z.sum_i = z.sum_xi = z.sum_yi = z.peak = 0ull;
                z.x_min = 0xFFFFFFFFu; z.x_max = 0u; z.y_min = 0xFFFFFFFFu; z.y_max = 0u;
                z.num_pixels = 0u; z.pad = 0u;
                s_acc[tid] = z;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kChainPer; ++q) {
                const uint32_t c = get_id(q) - c0;    // (an unused slot's 0xFFFF - c0 is never below kChainSlots: c0 < kChainLdsEntries)
                if (c < (uint32_t)kChainSlots) {
                    uint32_t yx = exy[q];
                    // (opaque: or the compiler computes every entry's products once, ahead of this loop over c0, and
                    // spills a hundred registers to hold them)
                    asm volatile("" : "+v"(yx));
                    const uint32_t y = yx >> 16, x = yx & 0xFFFFu;
                    const uint32_t ki = y * W + x;
                    const unsigned long long I = get_i(q);
                    ChainAcc* r = &s_acc[c];
                    atomicMin(&r->x_min, x); atomicMax(&r->x_max, x);
                    atomicMin(&r->y_min, y); atomicMax(&r->y_max, y);
                    atomicAdd(&r->num_pixels, 1u);
                    atomicAdd(&r->sum_i, I);
                    atomicAdd(&r->sum_xi, (2ull * x + 1ull) * I);
                    atomicAdd(&r->sum_yi, (2ull * y + 1ull) * I);
                    // highest intensity, ties -> smallest (y, x) = smallest k
                    // (connected_components.hpp:125-170, connected_components.cc:143-157)
                    atomicMax(&r->peak, (I << 32) | (unsigned long long)(0xFFFFFFFFu - ki));
                }
            }
            __syncthreads();
            const uint32_t here = min((uint32_t)kChainSlots, ncomp - c0);
            if ((uint32_t)tid < here) {
                const ChainAcc r = s_acc[tid];
                WireRec2 o;
                chain_record(sa, W, r.num_pixels, r.sum_i, r.sum_xi, r.sum_yi, r.x_min, r.x_max, r.y_min, r.y_max,
                             0xFFFFFFFFu - (uint32_t)(r.peak & 0xFFFFFFFFull), (uint32_t)(r.peak >> 32), s_sm, o);
                *reinterpret_cast<WireRec2*>(&s_out[tid * (sizeof(WireRec2) / 4)]) = o;
            }
            __syncthreads();
            {
                uint32_t* dst = reinterpret_cast<uint32_t*>(recs + c0);
                const uint32_t ndw = here * (uint32_t)(sizeof(WireRec2) / 4);
                for (uint32_t w = tid; w < ndw; w += kChainThreads) dst[w] = s_out[w];
            }
            __syncthreads();
        }
    } else if (runs) {
        // ---- denser frames of 16-bit pixels: the same phases over runs ------------------------------------------------------
        // s_row holds the list offset of every row's first RUN now.  A run lies inside one 32-pixel plane word = one aligned
        // 64-byte piece of its image row: phase P' fetches that piece with four 16-byte loads and sums the run in registers,
        // so the accumulators take one set of atomics per run instead of one per pixel.
        if constexpr (RUNS && sizeof(PixelT) == 2) {
        const uint32_t nr = s_row[a.H];
        uint32_t* s_rd = spar + kChainRunCap;   // descriptors: y << 19 | x0 << 5 | (len - 1)
        if (nr > (uint32_t)kChainRunCap) {
            run_flag = 16u;
        } else {
            // ---- X': descriptors into LDS in list order (wave after wave), the forest's singletons, the pixel values
            {
                const int tpw = (n_tiles + kChainWaves - 1) / kChainWaves;
                uint32_t base = 0;
                for (int v = 0; v < kChainWaves; ++v) {
                    const uint32_t cnt = s_wrun[v], src = s_toff[min(v * tpw, n_tiles)];
                    for (uint32_t j = tid; j < cnt; j += kChainThreads) {
                        const uint32_t i = base + j;
                        if (i < nr) {
                            s_rd[i] = src + j < a.cap ? gpar[src + j] : 0u;
                            spar[i] = i;
                        }
                    }
                    base += cnt;
                }
                // the intensity list (for the consumers of the pixel lists: nothing below reads it)
                if (a.need_lists)
                    for (uint32_t i = tid; i < n; i += kChainThreads) gi[i] = pixel_at(gk[i]);
            }
            __syncthreads();
            const uint32_t per = (nr + kChainThreads - 1) / kChainThreads;   // <= kChainRunPer
            const uint32_t i0 = min((uint32_t)tid * per, nr), i1 = min(i0 + per, nr);
            auto rx0 = [](uint32_t d) -> uint32_t { return (d >> 5) & 0x3FFFu; };
            auto rx1 = [](uint32_t d) -> uint32_t { return ((d >> 5) & 0x3FFFu) + (d & 31u); };

            // ---- U': the k + 1 edges between runs that touch (word boundaries; the reference's row-wrap edge), the k + W edges
            // between runs of consecutive rows that overlap
            for (uint32_t i = i0; i < i1; ++i) {
                const uint32_t d = s_rd[i], y = d >> 19, x0 = rx0(d), x1 = rx1(d);
                if (i > 0) {
                    const uint32_t dp = s_rd[i - 1], yp = dp >> 19, x1p = rx1(dp);
                    if ((yp == y && x1p + 1 == x0) || (x0 == 0 && yp + 1 == y && x1p == W - 1)) uf_union(spar, i - 1, i);
                }
                if (y + 1 >= H) continue;
                uint32_t lo = s_row[y + 1], hi = s_row[y + 2];
                const uint32_t end = hi;
                while (lo < hi) {   // the first run of the next row that ends at or after x0
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    if (rx1(s_rd[mid]) < x0) lo = mid + 1; else hi = mid;
                }
                for (uint32_t j = lo; j < end && rx0(s_rd[j]) <= x1; ++j) uf_union(spar, i, j);
            }
            __syncthreads();
            FFS_STOP_AFTER(A, 3);

            // ---- P': every run's sums (registers, through phase R'); roots, numbered in list order ----------------------
            uint32_t rd[kChainRunPer], rsi[kChainRunPer], rsj[kChainRunPer], rpk[kChainRunPer], rid[kChainRunPer];
            const rsrc_t r_img = make_rsrc(img, (uint32_t)a.H * a.pitch);
            uint32_t mine = 0;
#pragma unroll
            for (int q = 0; q < kChainRunPer; ++q) {
                rd[q] = 0; rsi[q] = 0; rsj[q] = 0; rpk[q] = 0; rid[q] = 0xFFFFu;
                if (i0 + q < i1) {
                    const uint32_t d = s_rd[i0 + q];
                    rd[q] = d;
                    const uint32_t y = d >> 19, x0 = rx0(d), len = (d & 31u) + 1u, b0 = x0 & 31u;
                    const uint32_t off = y * a.pitch + (x0 - b0) * 2u;   // the run's 64-byte piece (beyond the frame: zeros, never selected)
                    uint32_t v[16];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const u32x4 ld = __builtin_amdgcn_raw_buffer_load_b128(r_img, off, (uint32_t)t * 16u, 0);
                        v[4 * t] = ld[0]; v[4 * t + 1] = ld[1]; v[4 * t + 2] = ld[2]; v[4 * t + 3] = ld[3];
                    }
                    const uint32_t sel = (len == 32u ? 0xFFFFFFFFu : ((1u << len) - 1u)) << b0;
                    uint32_t si = 0, sj = 0, pk = 0;
#pragma unroll
                    for (int j = 0; j < 32; ++j) {
                        const uint32_t I = (j & 1) ? v[j >> 1] >> 16 : v[j >> 1] & 0xFFFFu;
                        const bool on = (sel >> j) & 1u;
                        const uint32_t Im = on ? I : 0u;
                        si += Im;
                        sj += (uint32_t)j * Im;                                   // < 32 * 32 * 65535
                        pk = max(pk, on ? (I << 5) | (uint32_t)(31 - j) : 0u);    // highest intensity, ties -> smallest x
                    }
                    rsi[q] = si;
                    rsj[q] = sj - b0 * si;   // sum (x - x0) I
                    rpk[q] = pk;
                    const uint32_t root = uf_find(spar, i0 + q);
                    rid[q] = root;
                    mine += root == i0 + q ? 1u : 0u;
                }
            }
            __syncthreads();   // every find is done: the forest's root slots now take the component numbers
            {
                uint32_t slot = block_exclusive_scan<kChainThreads>(mine, s_wave, before);
#pragma unroll
                for (int q = 0; q < kChainRunPer; ++q)
                    if (i0 + q < i1 && rid[q] == i0 + q) spar[i0 + q] = slot++;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kChainRunPer; ++q)
                if (i0 + q < i1) rid[q] = spar[rid[q]];
            __syncthreads();   // the forest and the descriptors are dead from here on: accumulators + record staging
            FFS_STOP_AFTER(A, 4);

            // ---- R': kChainSlots components at a time -------------------------------------------------------------------
            ChainAcc* s_acc = reinterpret_cast<ChainAcc*>(s_big);
            uint32_t* s_out = reinterpret_cast<uint32_t*>(s_big + kChainSlots * sizeof(ChainAcc));
            const uint32_t ncomp = min(before, sa.max_comp);
            for (uint32_t c0 = 0; c0 < ncomp; c0 += kChainSlots) {
                if (tid < kChainSlots) {
                    ChainAcc z;
                    z.sum_i = z.sum_xi = z.sum_yi = z.peak = 0ull;
                    z.x_min = 0xFFFFFFFFu; z.x_max = 0u; z.y_min = 0xFFFFFFFFu; z.y_max = 0u;
                    z.num_pixels = 0u; z.pad = 0u;
                    s_acc[tid] = z;
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < kChainRunPer; ++q) {
                    const uint32_t c = rid[q] - c0;    // (an unused slot's 0xFFFF - c0 is never below kChainSlots)
                    if (c < (uint32_t)kChainSlots) {
                        uint32_t d = rd[q];
                        asm volatile("" : "+v"(d));   // (see phase R: keeps the products inside the loop over c0)
                        const uint32_t y = d >> 19, x0 = rx0(d), x1 = rx1(d);
                        const unsigned long long si = rsi[q], sj = rsj[q];
                        const uint32_t pI = rpk[q] >> 5, px = (x0 & ~31u) + (31u - (rpk[q] & 31u));
                        ChainAcc* r = &s_acc[c];
                        atomicMin(&r->x_min, x0); atomicMax(&r->x_max, x1);
                        atomicMin(&r->y_min, y); atomicMax(&r->y_max, y);
                        atomicAdd(&r->num_pixels, x1 - x0 + 1u);
                        atomicAdd(&r->sum_i, si);
                        atomicAdd(&r->sum_xi, (2ull * x0 + 1ull) * si + 2ull * sj);   // sum (2x + 1) I over the run
                        atomicAdd(&r->sum_yi, (2ull * y + 1ull) * si);
                        atomicMax(&r->peak, ((unsigned long long)pI << 32) | (unsigned long long)(0xFFFFFFFFu - (y * W + px)));
                    }
                }
                __syncthreads();
                const uint32_t here = min((uint32_t)kChainSlots, ncomp - c0);
                if ((uint32_t)tid < here) {
                    const ChainAcc r = s_acc[tid];
                    WireRec2 o;
                    chain_record(sa, W, r.num_pixels, r.sum_i, r.sum_xi, r.sum_yi, r.x_min, r.x_max, r.y_min, r.y_max,
                                 0xFFFFFFFFu - (uint32_t)(r.peak & 0xFFFFFFFFull), (uint32_t)(r.peak >> 32), s_sm, o);
                    *reinterpret_cast<WireRec2*>(&s_out[tid * (sizeof(WireRec2) / 4)]) = o;
                }
                __syncthreads();
                {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(recs + c0);
                    const uint32_t ndw = here * (uint32_t)(sizeof(WireRec2) / 4);
                    for (uint32_t w = tid; w < ndw; w += kChainThreads) dst[w] = s_out[w];
                }
                __syncthreads();
            }
        }
        }
    } else {
        // ---- denser frames: the same stages on the global arrays (the bodies of k_union / k_reduce_roots / k_finalize_roots)
        for (uint32_t i = tid; i < n; i += kChainThreads) gi[i] = pixel_at(gk[i]);
        for (uint32_t i = tid; i < n; i += kChainThreads) {
            const uint32_t ki = gk[i];
            const uint32_t y = ki / W;
            const bool starts = i == 0 || gk[i - 1] + 1 != ki;
            if (!starts && ki - y * W == 0) uf_union(gpar, i - 1, i);
            if (y + 1 >= H) continue;
            uint32_t lo = max(i + 1, s_row[y + 1]), hi = min(min(n, i + 1 + W), s_row[y + 2]);
            const uint32_t key = ki + W;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (gk[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < n && gk[lo] == key) {
                if (starts || lo == 0 || gk[lo - 1] + 1 != key) uf_union(gpar, i, lo);
            }
        }
        __syncthreads();
        FFS_STOP_AFTER(A, 3);
        for (uint32_t i0 = tid; i0 < n; i0 += kChainThreads) {
            // the thread of a run's first entry (runs are also cut every 32 entries) sums the run in registers
            const uint32_t k0 = gk[i0];
            const uint32_t y = k0 / W, x0 = k0 - y * W;
            // (consecutive k across a row end -- the reference's row-wrap edge -- is one component but two rows: cut there)
            const bool owner = (i0 & 31u) == 0u || x0 == 0u || gk[i0 - 1] + 1 != k0;
            if (!owner) continue;
            const uint32_t ri = uf_find(gpar, i0);
            uint32_t npx = 0u, x_max = x0;
            unsigned long long s_i = 0, s_xi = 0, pk = 0;
            uint32_t j = i0;
            for (;;) {
                const unsigned long long I = gi[j];
                const uint32_t x = x0 + (j - i0);
                ++npx;
                x_max = x;
                s_i += I;
                s_xi += (2ull * x + 1ull) * I;
                pk = max(pk, (I << 32) | (unsigned long long)(0xFFFFFFFFu - j));   // ties -> smallest list index
                ++j;
                if (j >= n || (j & 31u) == 0u || x + 1 >= W || gk[j] != k0 + (j - i0)) break;
            }
            CompAcc2* r = gacc + ri;
            atomicMin(&r->x_min, x0); atomicMax(&r->x_max, x_max);
            atomicMin(&r->y_min, y); atomicMax(&r->y_max, y);
            atomicAdd(&r->num_pixels, npx);
            atomicAdd(&r->sum_i, s_i);
            atomicAdd(&r->sum_xi, s_xi);
            atomicAdd(&r->sum_yi, (2ull * y + 1ull) * s_i);
            atomicMax(&r->peak, pk);
        }
        __threadfence();
        __syncthreads();
        uint32_t* s_out = reinterpret_cast<uint32_t*>(s_big);
        for (uint32_t base = 0; base < n; base += kChainThreads) {
            const uint32_t i = base + (uint32_t)tid;
            const bool root = i < n && ld_parent(gpar + i) == i;
            uint32_t nroots;
            const uint32_t slot = block_exclusive_scan<kChainThreads>(root ? 1u : 0u, s_wave, nroots);
            if (root && before + slot < sa.max_comp) {
                // agent-scope loads: the accumulator was built by atomics at the L2
                const unsigned long long* p = reinterpret_cast<const unsigned long long*>(gacc + i);
                unsigned long long q[6];
#pragma unroll
                for (int w = 0; w < 6; ++w) q[w] = __hip_atomic_load(p + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t npx = __hip_atomic_load(&(gacc + i)->num_pixels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t pi = min(0xFFFFFFFFu - (uint32_t)(q[3] & 0xFFFFFFFFull), n - 1);  // (never clamps: every root owns a pixel)
                WireRec2 o;
                chain_record(sa, W, npx, q[0], q[1], q[2], (uint32_t)q[4], (uint32_t)(q[4] >> 32), (uint32_t)q[5], (uint32_t)(q[5] >> 32),
                             gk[pi], (uint32_t)(q[3] >> 32), s_sm, o);
                *reinterpret_cast<WireRec2*>(&s_out[slot * (sizeof(WireRec2) / 4)]) = o;
            }
            __syncthreads();
            {
                const uint32_t first = min(before, sa.max_comp), last = min(before + nroots, sa.max_comp);
                uint32_t* dst = reinterpret_cast<uint32_t*>(recs + first);
                const uint32_t ndw = (last - first) * (uint32_t)(sizeof(WireRec2) / 4);
                for (uint32_t w = tid; w < ndw; w += kChainThreads) dst[w] = s_out[w];
            }
            before += nroots;
            __syncthreads();
        }
    }

    // ---- counters: device copies for the other consumers of the lists, host copies for ffs_wait() --------------
    __syncthreads();
    if (tid == 0) {
        uint32_t flags = *a.overflow | bright_flag | run_flag;   // what the dense stages raised (corrupt chunk; bright-list overflow)
        if (total > a.cap) flags |= 1u;
        if (before > sa.max_comp) flags |= 2u;
        a.num_strong[frame] = total;
        a.n_comp[frame] = before;
        const size_t B = A.max_batch;
        FFS_PHASE_TS(A, 6);
        A.h_counts[frame] = total;
        A.h_counts[B + frame] = before;
        A.h_counts[10 * B + 1 + frame] = flags;
    }
    if (tid < 8) {
        a.summary[(uint64_t)frame * 8 + tid] = s_sm[tid];
        A.h_counts[2 * (size_t)A.max_batch + (size_t)frame * 8 + tid] = s_sm[tid];
    }
}
template __global__ void k_frame_chain<uint16_t, false>(const ChainArgs);
template __global__ void k_frame_chain<uint16_t, true>(const ChainArgs);
template __global__ void k_frame_chain<uint16_t, false, true>(const ChainArgs);
template __global__ void k_frame_chain<uint32_t, false, true>(const ChainArgs);
template __global__ void k_frame_chain<uint32_t, false>(const ChainArgs);

}  // namespace ffsamd
