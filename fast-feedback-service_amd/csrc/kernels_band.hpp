// kernels_band.hpp (included by ffs_submit.hip) -- the sparse stage of the standard path in SMALL workgroups (round 5).
//
// Why: k_frame_chain gives a frame one workgroup of 1024 threads and 156 KB of LDS -- a whole CU for ~170 us.  It can only start
// on a CU that has drained completely, and no streaming wave can share the CU with it: with 48-56 frames per batch the streaming
// kernel alone takes 8.8-8.9 us per frame instead of 9.7, and the step stayed at 9.9-10.1 because of those 32-56 CUs
// (profiles/r04l_batch_size_ab.txt, r04o_chain_phase_times.txt).
//
// Here the same work is cut along the BANDS of the streaming kernel (a band = the ~78 rows one streaming wave marches down):
//   k_band_cc      one WAVE per (frame, band): merges the wave logs of the band's strips into the band's strong pixels in raster
//                  order (decides the undecided bright-window pixels on the way), labels them with a union-find forest in ~11 KB of
//                  LDS, and leaves per band: the accumulators of its components (in order of their first pixel), and the strong
//                  pixels of its first and last row with their component -- the seams
//   k_frame_merge  one workgroup of 256 threads per frame: joins the bands' components across the seams (the k + W edges between a
//                  band's last row and the next band's first, and the reference's row-wrap edge between them,
//                  connected_components.cc:62-70), numbers the components in order of their first pixel (= label order, :91,242),
//                  folds the accumulators of joined components, and writes records, counters and flags exactly as k_frame_chain does
// A band wave fits wherever one streaming wave has left (64 threads, <= 64 VGPRs, 11 KB of LDS), so the launch neither waits for
// CUs to drain nor keeps streaming waves off them.  Results are k_frame_chain's bit for bit: same edges, same integer
// accumulators, same order, the same chain_record().
//
// Scope: batches whose strong pixels travel in wave logs and whose lists nobody reads (need_lists == 0, no byte mask) -- the hot
// path's default.  A band the plan below cannot hold (more than kBandPx strong pixels, kBandEntries log entries, kBandCompStride
// components; a seam row with more than kBandSeamCap strong pixels) raises flag 128: ffs_wait runs the batch again through
// k_frame_chain and the stream stays there for a while (dense data).
#pragma once
#include "kernels_chain.hpp"

namespace ffsamd {

constexpr int kBandMaxRows = 128;       // rows of a band (the streaming kernel's geometry: 72-80 for Eiger frames)
constexpr int kBandChunks = 8;          // chunks of 64 log entries held in registers
constexpr int kBandEntries = 64 * kBandChunks;
constexpr int kBandPx = 768;            // strong pixels of one band of one frame (the bench frames: 320 +- 70)
constexpr int kBandPer = kBandPx / 64;  // consecutive list entries per lane in the forest phases
constexpr int kBandCw = 1024;           // (row, strip) counters: rows * strips of the frame
constexpr int kBandSlots = 64;          // components accumulated in LDS at a time
constexpr int kBandCompStride = 256;    // components of a band handed to the merge (accumulator slots per band)
constexpr int kBandSeamCap = 256;       // strong pixels of a band's first / last row handed to the merge
constexpr int kBandItems = (kBandPx - kBandEntries) / 2;   // undecided pixels of a band: two words each behind the per-entry results, in the forest's LDS (free then)
constexpr int kMergeThreads = 256;
constexpr int kMergeMaxBands = 128;
constexpr int kMergeCap = 4096;         // band components of a frame whose forest fits the merge's LDS
constexpr int kMergeSeamLds = 8;        // strong pixels of a seam row the merge stages in LDS (longer rows: walked in global memory)
constexpr int kMergeLabels = 2;         // labels per thread and round of the record phase
constexpr int kMergePer = kMergeCap / kMergeThreads;
static_assert(sizeof(ChainAcc) == 56 && kBandSlots * (int)sizeof(ChainAcc) <= kBandCw * 4, "the accumulators take the counters' LDS");
static_assert(kBandEntries + 2 * kBandItems <= kBandPx && kBandMaxRows % 2 == 0, "LDS plan");

constexpr int kBandSplitRows = 96;      // a streaming band taller than this is cut into sub-bands of equal height, a wave each
struct BandArgs {
    ChainArgs A;
    int sub, sub_rows;   // sub-bands per streaming band (1: the band itself), rows of each (large batches have taller bands: fewer, longer
                         // streaming waves; the band waves' LDS plan stays with ~80 rows).  "Band" below = sub-band.
    uint4* hdr;          // [frames][n_bands * sub] strong pixels, components, first-row pixels | last-row pixels << 16, flags | rows << 16
    ChainAcc* acc;       // [frames][n_bands][kBandCompStride]
    uint32_t* seam;      // [frames][n_bands][2][kBandSeamCap]  x | component << 16  (first row, last row)
};

template <typename PixelT>
__global__ __launch_bounds__(64) void k_band_cc(const BandArgs B) {
    const ThresholdArgs& T = B.A.t;
    const CclArgs& a = B.A.c;
    if (T.dbg_prio & 2) __builtin_amdgcn_s_setprio(3);   // (tuning "stream_prio" bit 2: measured, not the default)
    __shared__ uint32_t s_lst[16];
    __shared__ uint32_t s_row[kBandMaxRows + 2];
    __shared__ __align__(16) uint32_t s_cw[kBandCw];
    __shared__ __align__(8) uint2 s_ent[kBandEntries];   // the band's log entries: (row << 16 | group, strip << 16 | undecided << 8 | strong); not mine: y = 0xFFFF0000
    __shared__ uint16_t s_x[kBandPx];
    __shared__ uint8_t s_y[kBandPx];   // the pixel's row in the band (the loops below ask for it instead of walking the rows' offsets)
    __shared__ PixelT s_i[kBandPx];
    __shared__ uint32_t s_par[kBandPx];
    __shared__ uint32_t s_flag;

    const int band = (int)blockIdx.x / B.sub, sub = (int)blockIdx.x - band * B.sub, frame = blockIdx.y, lane = threadIdx.x;   // band: the streaming kernel's
    const uint32_t W = (uint32_t)a.W;
    constexpr uint32_t kGroupPx = sizeof(PixelT) == 2 ? 8u : 4u;
    // where this frame's groups lie in the streaming launch (see k_frame_chain, phase L1)
    const uint32_t l_y = (uint32_t)frame / (uint32_t)T.group_frames;
    const uint32_t l_fe = (uint32_t)frame - l_y * (uint32_t)T.group_frames;
    const uint32_t gsep = (uint32_t)T.gpf + 1u, G0 = l_fe * gsep, G1 = G0 + (uint32_t)T.gpf - 1u;
    const uint32_t l_s0 = G0 / (uint32_t)kSOwned;
    const uint32_t l_ns = min(G1 / (uint32_t)kSOwned - l_s0 + 1u, 16u);
    const uint32_t wave0 = log_slot(T, l_y, (uint32_t)band, l_s0);   // strip k: + k
    const int sb1 = min(band_first_row(band + 1, T.band_rows, T.band_rows2, T.band_split), a.H);   // end of the streaming band
    const int yb0 = min(band_first_row(band, T.band_rows, T.band_rows2, T.band_split) + sub * B.sub_rows, sb1);
    const int yb1 = min(yb0 + B.sub_rows, sb1);
    const uint32_t rows = (uint32_t)(yb1 - yb0);   // (0: a short last band has no such sub-band)
    const uint8_t* img = (const uint8_t*)a.image + (uint64_t)frame * a.frame_stride;
    const size_t slot = (size_t)frame * (size_t)gridDim.x + (size_t)blockIdx.x;

    FFS_STOP_AFTER(B.A, 10);   // (timing experiments: 10 = both kernels return at once, 11 = the merge does, 12 / 13 = the band wave after its entries / the placement)
    if (lane == 0) s_flag = (rows > (uint32_t)kBandMaxRows || rows * l_ns > (uint32_t)kBandCw) ? 128u : 0u;   // (the host checks the geometry too)
    for (uint32_t i = lane; i < min(rows * l_ns, (uint32_t)kBandCw); i += 64) s_cw[i] = 0;
    for (uint32_t i = lane; i < kBandEntries; i += 64) s_par[i] = 0;   // (phase M's per-entry results)
    __syncthreads();

    // ---- the logs of the band's strips, laid end to end ------------------------------------------------------------------------
    uint32_t cnt = 0;
    if ((uint32_t)lane < l_ns && rows != 0) {
        cnt = T.wlog_n[wave0 + (uint32_t)lane];
        if (cnt > (uint32_t)kWlogCap) { atomicOr(&s_flag, 32u); cnt = (uint32_t)kWlogCap; }
    }
    const uint32_t incl0 = wave_inclusive_scan(cnt);
    if (lane < 16) s_lst[lane] = incl0 - cnt;
    const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
    if (nb > (uint32_t)kBandEntries) atomicOr(&s_flag, 128u);
    __syncthreads();
    uint32_t n = 0, ncomp = 0, n_top = 0, n_bot = 0;
    if (s_flag == 0 && nb != 0) {
        // the entries into LDS (all loads issued before any is used); what is not this frame's, or another sub-band's row, is nobody's
        {
            uint2 ent[kBandChunks];
#pragma unroll
            for (int c = 0; c < kBandChunks; ++c) {
                const uint32_t f = (uint32_t)(c * 64 + lane);
                ent[c] = make_uint2(0xFFFFFFFFu, 0xFFFF0000u);
                if (f < nb) {
                    uint32_t k = 0;
                    for (uint32_t j = 1; j < l_ns; ++j) k += f >= s_lst[j] ? 1u : 0u;
                    ent[c] = T.wlog[(uint64_t)(wave0 + k) * kWlogCap + (f - s_lst[k])];
                    ent[c].y = ((ent[c].y >> 16) != l_fe || (int)(ent[c].x >> 16) < yb0 || (int)(ent[c].x >> 16) >= yb1) ? 0xFFFF0000u : (k << 16) | (ent[c].y & 0xFFFFu);
                }
            }
#pragma unroll
            for (int c = 0; c < kBandChunks; ++c)
                if ((uint32_t)(c * 64) < nb) s_ent[c * 64 + lane] = ent[c];
        }
        __syncthreads();
        FFS_STOP_AFTER(B.A, 12);
        const uint32_t nchunks = (nb + 63u) / 64u;
        auto mine_of = [](uint2 v) -> bool { return (v.y >> 16) != 0xFFFFu; };
        // the logged pixels of an entry that holds a strong or undecided pixel (phase L places them); the first two chunks' are asked
        // for here, ahead of the gathers of phase M, the others one chunk ahead of their use
        auto load_px = [&](uint32_t c) -> uint4 {
            uint4 px = make_uint4(0u, 0u, 0u, 0u);
            if (c < nchunks) {
                const uint32_t f = c * 64u + (uint32_t)lane;
                const uint2 v = s_ent[f];
                if (f < nb && mine_of(v) && (v.y & 0xFFFFu)) px = T.wpix[(uint64_t)(wave0 + (v.y >> 16)) * kWlogCap + (f - s_lst[v.y >> 16])];
            }
            return px;
        };
        const uint4 px0 = load_px(0), px1 = load_px(1);

        // ---- M: undecided pixels (windows with sum p >= 65536: the cores of bright spots) take the gathered predicate, one per lane --
        {
            // per-entry results in s_par[0 .. kBandEntries) (zeroed above), the items behind them
            uint32_t* items = s_par + kBandEntries;
            uint32_t nm = 0;   // wave-uniform
            for (uint32_t c = 0; c < nchunks; ++c) {
                const uint2 v = s_ent[c * 64 + lane];
                uint32_t maybe = mine_of(v) ? (v.y >> 8) & 0xFFu & ~v.y : 0u;
                const uint32_t pc = (uint32_t)__popc(maybe);
                const uint32_t inc = wave_inclusive_scan(pc);
                uint32_t q = nm + inc - pc;
                nm += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                while (maybe) {
                    const int b = __ffs((int)maybe) - 1;
                    maybe &= maybe - 1;
                    if (q < (uint32_t)kBandItems) {
                        items[2 * q] = v.x;                                                  // row << 16 | group
                        items[2 * q + 1] = ((c * 64u + (uint32_t)lane) << 3) | (uint32_t)b;   // entry slot, bit
                    }
                    ++q;
                }
            }
            if (nm > (uint32_t)kBandItems) atomicOr(&s_flag, 128u);
            if (nm != 0) {   // wave-uniform
                __syncthreads();
                for (uint32_t i = lane; i < min(nm, (uint32_t)kBandItems); i += 64) {
                    const uint32_t rg = items[2 * i], sb = items[2 * i + 1];
                    const uint32_t row = rg >> 16, ge = rg & 0xFFFFu, b = sb & 7u;
                    if (exact_strong_lite<PixelT>(T, img, (int)(ge * kGroupPx + b), (int)row)) atomicOr(&s_par[sb >> 3], 1u << b);
                }
                __syncthreads();
                for (uint32_t f = lane; f < nb; f += 64) s_ent[f].y |= s_par[f];   // (zero in everybody else's slots)
                __syncthreads();
            }
        }

        // ---- C: strong pixels per (row, strip), rows' list offsets --------------------------------------------------------------
        for (uint32_t f = lane; f < nb; f += 64) {
            const uint2 v = s_ent[f];
            const uint32_t pc = (uint32_t)__popc(v.y & 0xFFu);
            if (mine_of(v) && pc) atomicAdd(&s_cw[((v.x >> 16) - (uint32_t)yb0) * l_ns + (v.y >> 16)], pc);
        }
        __syncthreads();
        {
            // lane L owns rows 2 L and 2 L + 1: counts -> positions (the row's own offset + the strips before)
            const uint32_t r0 = 2u * (uint32_t)lane, r1 = r0 + 1u;
            uint32_t t0 = 0, t1 = 0;
            if (r0 < rows) for (uint32_t k = 0; k < l_ns; ++k) t0 += s_cw[r0 * l_ns + k];
            if (r1 < rows) for (uint32_t k = 0; k < l_ns; ++k) t1 += s_cw[r1 * l_ns + k];
            const uint32_t inc = wave_inclusive_scan(t0 + t1);
            n = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            uint32_t at = inc - (t0 + t1);
            if (r0 < rows) {
                s_row[r0] = at;
                for (uint32_t k = 0; k < l_ns; ++k) { const uint32_t c = s_cw[r0 * l_ns + k]; s_cw[r0 * l_ns + k] = at; at += c; }
            }
            if (r1 < rows) {
                s_row[r1] = at;
                for (uint32_t k = 0; k < l_ns; ++k) { const uint32_t c = s_cw[r1 * l_ns + k]; s_cw[r1 * l_ns + k] = at; at += c; }
            }
            if (lane == 0) { s_row[rows] = n; s_row[rows + 1] = n; }
        }
        if (n > (uint32_t)kBandPx) atomicOr(&s_flag, 128u);
        __syncthreads();

        if (s_flag == 0 && n != 0) {
            // ---- L: every entry's pixels placed in raster order (k_frame_chain's phase L2, one band) ------------------------------
            uint4 px_next = px0;
            for (uint32_t c = 0; c < nchunks; ++c) {
                const uint4 px = px_next;
                px_next = c == 0 ? px1 : load_px(c + 1);   // (on its way while this chunk is placed)
                const uint2 v = s_ent[c * 64 + lane];
                const bool mine = mine_of(v);
                const uint32_t row = (v.x >> 16) - (uint32_t)yb0, k = v.y >> 16;
                const uint32_t cb = mine ? v.y & 0xFFu : 0u;
                const uint32_t pc = (uint32_t)__popc(cb);
                const uint32_t incl = wave_inclusive_scan(pc), excl = incl - pc;
                // the pixels of MY (strip, row) before me in this chunk: a log is sorted by row and the logs follow each other, so a
                // (strip, row) is a run of lanes; lanes that are not mine carry keys of their own and no pixels
                const uint32_t key = mine ? (k << 16) | row : 0x80000000u + (uint32_t)lane;
                const uint32_t key_prev = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)key, 0x138, 0xf, 0xf, false);   // wave_shr:1
                const bool starts = lane == 0 || key_prev != key;
                const uint32_t seg = wave_inclusive_max(starts ? excl : 0u);
                if (mine && cb) {
                    uint32_t at = s_cw[row * l_ns + k] + (excl - seg);
                    const uint32_t ge = v.x & 0xFFFFu;
                    uint32_t w = cb;
                    while (w) {
                        const int b = __ffs((int)w) - 1;
                        w &= w - 1;
                        uint32_t I;
                        if constexpr (sizeof(PixelT) == 2) {
                            const uint32_t pw = b < 2 ? px.x : b < 4 ? px.y : b < 6 ? px.z : px.w;
                            I = (b & 1) ? pw >> 16 : pw & 0xFFFFu;
                        } else {
                            I = b == 0 ? px.x : b == 1 ? px.y : b == 2 ? px.z : px.w;
                        }
                        if (at < (uint32_t)kBandPx) {
                            s_x[at] = (uint16_t)(ge * kGroupPx + (uint32_t)b);
                            s_i[at] = (PixelT)I;
                            s_y[at] = (uint8_t)row;
                        }
                        ++at;
                    }
                }
                __syncthreads();
                // the next chunk may continue the chunk's last segment: move that (row, strip)'s position on
                const uint32_t key_next = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)key, 0x130, 0xf, 0xf, false);   // wave_shl:1
                if (mine && (lane == 63 || key_next != key)) s_cw[row * l_ns + k] += incl - seg;
                __syncthreads();
            }

            FFS_STOP_AFTER(B.A, 13);
            // ---- X / U: the forest (k_frame_chain's phases, band-local rows; edges across the band's ends are the merge's) --------
            // (A lane takes `per` consecutive pixels; the loops run `per` times for the whole wave with the lane's share as a
            // predicate -- a scalar counter instead of an exec-mask loop per lane -- and a pixel's row comes from s_y: round 5 measured
            // what this half of the wave costs the streaming waves beside it)
            const uint32_t per = (n + 63u) / 64u;
            const uint32_t i0 = min((uint32_t)lane * per, n), i1 = min(i0 + per, n);
            // pixel j continues the run of pixel j - 1: the k + 1 edge, the reference's row-wrap included (x = W - 1 of one row and
            // x = 0 of the next: connected_components.cc:62-70); (xp, yp) = pixel j - 1
            auto continues = [&](uint32_t j, uint32_t xj, uint32_t yj, uint32_t xp, uint32_t yp) -> bool {
                return j != 0 && (yj == yp ? xp + 1 == xj : (xj == 0 && xp == W - 1 && yp + 1 == yj));
            };
            const uint32_t xp0 = i0 > 0 && i0 < i1 ? (uint32_t)s_x[i0 - 1] : 0u, yp0 = i0 > 0 && i0 < i1 ? (uint32_t)s_y[i0 - 1] : 0u;
            {
                uint32_t xp = xp0, yp = yp0;
                for (uint32_t k = 0; k < per; ++k) {
                    const uint32_t i = i0 + k;
                    if (i < i1) {
                        const uint32_t x = s_x[i], y = s_y[i];
                        s_par[i] = (continues(i, x, y, xp, yp) && x != 0) ? i - 1 : i;
                        xp = x; yp = y;
                    }
                }
            }
            __syncthreads();
            {
                uint32_t xp = xp0, yp = yp0;
                for (uint32_t k = 0; k < per; ++k) {
                    const uint32_t i = i0 + k;
                    if (i < i1) {
                        const uint32_t x = s_x[i], y = s_y[i];
                        const bool cont = continues(i, x, y, xp, yp);
                        xp = x; yp = y;
                        if (cont && x == 0) uf_union(s_par, i - 1, i);   // the row-wrap edge inside the band
                        if (y + 1 < rows) {
                            uint32_t lo = max(i + 1, s_row[y + 1]);
                            const uint32_t end = max(lo, s_row[y + 2]);
                            uint32_t hi = end;
                            while (lo < hi) {   // lower bound of x among the next row's entries
                                const uint32_t mid = lo + ((hi - lo) >> 1);
                                if ((uint32_t)s_x[mid] < x) lo = mid + 1; else hi = mid;
                            }
                            if (lo < end && (uint32_t)s_x[lo] == x) {
                                // one edge per pair of overlapping runs is enough
                                if (!cont || !continues(lo, x, y + 1, s_x[lo - 1], s_y[lo - 1])) uf_union(s_par, i, lo);
                            }
                        }
                    }
                }
            }
            __syncthreads();

            // ---- P: roots, numbered in list order; every entry's component (16 bits each, in the entries' LDS: they are done with) ----
            uint16_t* s_id = reinterpret_cast<uint16_t*>(s_ent);
            uint32_t mine_roots = 0;
            for (uint32_t k = 0; k < per; ++k) {
                const uint32_t i = i0 + k;
                if (i < i1) {
                    const uint32_t root = uf_find(s_par, i);
                    s_id[i] = (uint16_t)root;
                    mine_roots += root == i ? 1u : 0u;
                }
            }
            __syncthreads();   // every find is done: the root slots take the component numbers
            {
                const uint32_t inc = wave_inclusive_scan(mine_roots);
                ncomp = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                uint32_t sl = inc - mine_roots;
                for (uint32_t k = 0; k < per; ++k) {
                    const uint32_t i = i0 + k;
                    if (i < i1 && (uint32_t)s_id[i] == i) s_par[i] = sl++;
                }
            }
            __syncthreads();
            for (uint32_t k = 0; k < per; ++k) {
                const uint32_t i = i0 + k;
                if (i < i1) s_id[i] = (uint16_t)s_par[s_id[i]];
            }
            if (ncomp > (uint32_t)kBandCompStride) atomicOr(&s_flag, 128u);
            __syncthreads();

            if (s_flag == 0) {
                // ---- R: kBandSlots components at a time into LDS accumulators, then out ------------------------------------------
                ChainAcc* s_acc = reinterpret_cast<ChainAcc*>(s_cw);
                ChainAcc* out = B.acc + slot * kBandCompStride;
                for (uint32_t c0 = 0; c0 < ncomp; c0 += kBandSlots) {
                    {
                        ChainAcc z;
                        z.sum_i = z.sum_xi = z.sum_yi = z.peak = 0ull;
                        z.x_min = 0xFFFFFFFFu; z.x_max = 0u; z.y_min = 0xFFFFFFFFu; z.y_max = 0u;
                        z.num_pixels = 0u; z.pad = 0u;
                        s_acc[lane] = z;
                    }
                    __syncthreads();
                    {
                        for (uint32_t k = 0; k < per; ++k) {
                            const uint32_t i = i0 + k;
                            const uint32_t c = i < i1 ? (uint32_t)s_id[i] - c0 : 0xFFFFFFFFu;
                            if (c < (uint32_t)kBandSlots) {
                                const uint32_t y = (uint32_t)yb0 + (uint32_t)s_y[i], x = s_x[i];
                                const uint32_t ki = y * W + x;
                                const unsigned long long I = s_i[i];
                                ChainAcc* r = &s_acc[c];
                                atomicMin(&r->x_min, x); atomicMax(&r->x_max, x);
                                atomicMin(&r->y_min, y); atomicMax(&r->y_max, y);
                                atomicAdd(&r->num_pixels, 1u);
                                atomicAdd(&r->sum_i, I);
                                atomicAdd(&r->sum_xi, (2ull * x + 1ull) * I);
                                atomicAdd(&r->sum_yi, (2ull * y + 1ull) * I);
                                // highest intensity, ties -> smallest (y, x) = smallest k (connected_components.hpp:125-170, .cc:143-157)
                                atomicMax(&r->peak, (I << 32) | (unsigned long long)(0xFFFFFFFFu - ki));
                            }
                        }
                    }
                    __syncthreads();
                    {
                        const uint32_t here = min((uint32_t)kBandSlots, ncomp - c0);
                        const uint32_t ndw = here * (uint32_t)(sizeof(ChainAcc) / 4);
                        uint32_t* dst = reinterpret_cast<uint32_t*>(out + c0);
                        const uint32_t* src = reinterpret_cast<const uint32_t*>(s_acc);
                        for (uint32_t w = lane; w < ndw; w += 64) dst[w] = src[w];
                    }
                    __syncthreads();
                }
                // ---- the seams: the strong pixels of the band's first and last row, with their component ----------------------------
                n_top = s_row[1] - s_row[0];
                n_bot = s_row[rows] - s_row[rows - 1];
                if (n_top > (uint32_t)kBandSeamCap || n_bot > (uint32_t)kBandSeamCap) atomicOr(&s_flag, 128u);
                uint32_t* seam = B.seam + slot * 2 * kBandSeamCap;
                for (uint32_t j = lane; j < min(n_top, (uint32_t)kBandSeamCap); j += 64) seam[j] = (uint32_t)s_x[j] | ((uint32_t)s_id[j] << 16);
                const uint32_t bb = s_row[rows - 1];
                for (uint32_t j = lane; j < min(n_bot, (uint32_t)kBandSeamCap); j += 64) seam[kBandSeamCap + j] = (uint32_t)s_x[bb + j] | ((uint32_t)s_id[bb + j] << 16);
            }
        }
    }
    __syncthreads();
    if (lane == 0) B.hdr[slot] = make_uint4(n, ncomp, n_top | (n_bot << 16), s_flag | (rows << 16));
}
template __global__ void k_band_cc<uint16_t>(const BandArgs);
template __global__ void k_band_cc<uint32_t>(const BandArgs);

// One workgroup per frame: the bands' components joined across the seams, numbered, folded, written.
__global__ __launch_bounds__(kMergeThreads) void k_frame_merge(const BandArgs B) {
    const ChainArgs& A = B.A;
    const CclArgs& a = A.c;
    const SegArgs& sa = A.s;
    const ThresholdArgs& T = A.t;
    if (T.dbg_prio & 2) __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t s_off[kMergeMaxBands + 1];
    __shared__ uint32_t s_par[kMergeCap];
    __shared__ uint16_t s_l2id[kMergeCap];          // label -> band component (the root with that label)
    __shared__ uint32_t s_touched[kMergeCap / 32];  // roots that members were folded into (their accumulators were changed by atomics)
    __shared__ uint32_t s_wave[kMergeThreads / 64];
    __shared__ uint32_t s_sm[8];
    __shared__ uint32_t s_tot[2];   // strong pixels, flags
    __shared__ __align__(16) uint32_t s_out[kMergeLabels * kMergeThreads * (sizeof(WireRec2) / 4)];
    __shared__ __align__(16) uint32_t s_seam[kMergeMaxBands * 2 * kMergeSeamLds];   // the first pixels of every seam row (first row, last row per band)
    const int frame = blockIdx.x, tid = threadIdx.x;
    const uint32_t W = (uint32_t)a.W;
    const uint32_t nbands = (uint32_t)(T.n_bands * B.sub);
    const uint4* hdr = B.hdr + (size_t)frame * nbands;
    ChainAcc* acc = B.acc + (size_t)frame * nbands * kBandCompStride;
    const uint32_t* seam = B.seam + (size_t)frame * nbands * 2 * kBandSeamCap;

    FFS_PHASE_TS(A, 0);
    FFS_STOP_AFTER(A, 10);
    FFS_STOP_AFTER(A, 11);
    FFS_STOP_AFTER(A, 12);
    FFS_STOP_AFTER(A, 13);
    if (tid < 8) s_sm[tid] = 0;
    if (tid < 2) s_tot[tid] = 0;
    for (int i = tid; i < kMergeCap / 32; i += kMergeThreads) s_touched[i] = 0;
    __syncthreads();
    uint4 h = make_uint4(0u, 0u, 0u, 0u);
    if ((uint32_t)tid < nbands) h = hdr[tid];
    // the seam rows' first pixels into LDS, all loads in one round trip (the walk below then costs no memory access per step)
    for (uint32_t j = (uint32_t)tid; j < nbands * 2u * (kMergeSeamLds / 4); j += kMergeThreads) {
        const uint32_t list = j / (kMergeSeamLds / 4), q = j % (kMergeSeamLds / 4);
        reinterpret_cast<uint4*>(s_seam)[j] = reinterpret_cast<const uint4*>(seam + (size_t)list * kBandSeamCap)[q];
    }
    if (h.x) atomicAdd(&s_tot[0], h.x);
    if (h.w & 0xFFFFu) atomicOr(&s_tot[1], h.w & 0xFFFFu);
    uint32_t NB;
    const uint32_t my_off = block_exclusive_scan<kMergeThreads>(h.y, s_wave, NB);
    if ((uint32_t)tid < nbands) s_off[tid] = my_off;
    if (tid == 0) s_off[nbands] = NB;
    __syncthreads();
    const uint32_t total = s_tot[0];
    uint32_t flags = s_tot[1];
    if (NB > (uint32_t)kMergeCap) flags |= 128u;
    FFS_PHASE_TS(A, 1);
    uint32_t before = 0;

    if (flags == 0) {
        const uint32_t per = (NB + kMergeThreads - 1) / kMergeThreads;   // <= kMergePer
        const uint32_t i0 = min((uint32_t)tid * per, NB), i1 = min(i0 + per, NB);
        for (uint32_t i = i0; i < i1; ++i) s_par[i] = i;
        __syncthreads();
        // ---- the seams: band b's last row against the first row of the band below it (consecutive image rows) -----------------------
        if ((uint32_t)tid + 1 < nbands && (h.z >> 16) != 0u) {
            const uint32_t b = (uint32_t)tid;
            uint32_t t = b + 1;   // the band below: the next one that has rows at all (a short last streaming band has no second sub-band)
            uint4 ht = hdr[t];
            while ((ht.w >> 16) == 0u && t + 1 < nbands) ht = hdr[++t];
            const uint32_t nbot = h.z >> 16, ntop = (ht.w >> 16) != 0u ? ht.z & 0xFFFFu : 0u;
            if (nbot && ntop) {
                const uint32_t* bot = seam + ((size_t)b * 2 + 1) * kBandSeamCap;
                const uint32_t* top = seam + ((size_t)t * 2) * kBandSeamCap;
                const uint32_t* lbot = s_seam + (b * 2u + 1u) * kMergeSeamLds;
                const uint32_t* ltop = s_seam + (t * 2u) * kMergeSeamLds;
                auto bot_at = [&](uint32_t i) -> uint32_t { return i < (uint32_t)kMergeSeamLds ? lbot[i] : bot[i]; };
                auto top_at = [&](uint32_t i) -> uint32_t { return i < (uint32_t)kMergeSeamLds ? ltop[i] : top[i]; };
                const uint32_t ob = s_off[b], ot = s_off[t];
                // the reference's k + 1 edge has no row-end check (connected_components.cc:62-70): the last pixel of a row and the
                // first of the next are joined when both are strong
                const uint32_t last = bot_at(nbot - 1), first = top_at(0);
                if ((last & 0xFFFFu) == W - 1 && (first & 0xFFFFu) == 0u) uf_union(s_par, ob + (last >> 16), ot + (first >> 16));
                uint32_t p = 0, q = 0;   // the k + W edges: equal x (both lists ascend)
                uint32_t vb = bot_at(0), vt = first;
                for (;;) {
                    const uint32_t xb = vb & 0xFFFFu, xt = vt & 0xFFFFu;
                    if (xb == xt) uf_union(s_par, ob + (vb >> 16), ot + (vt >> 16));
                    if (xb <= xt) { if (++p >= nbot) break; vb = bot_at(p); }
                    else { if (++q >= ntop) break; vt = top_at(q); }
                }
            }
        }
        __syncthreads();
        FFS_PHASE_TS(A, 2);
        // ---- roots, numbered in order (band after band, inside a band by first pixel: the order of the components' first pixels) ----
        // a band component's accumulator: the band of id i by search in s_off (ids ascend band after band)
        auto acc_of = [&](uint32_t id) -> ChainAcc* {
            uint32_t lo = 0, hi = nbands - 1;   // the last band whose offset is <= id (bands without components share their successor's)
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo + 1) >> 1);
                if (s_off[mid] <= id) lo = mid; else hi = mid - 1;
            }
            return acc + (size_t)lo * kBandCompStride + (id - s_off[lo]);
        };
        uint32_t mine = 0;
        for (uint32_t i = i0; i < i1; ++i) {
            const uint32_t root = uf_find(s_par, i);
            if (root == i) {
                ++mine;
            } else {
                // a joined component: its accumulator folded into its root's (global atomics: few, and all from this block)
                atomicOr(&s_touched[root >> 5], 1u << (root & 31u));
                const ChainAcc m = *acc_of(i);
                ChainAcc* r = acc_of(root);
                atomicMin(&r->x_min, m.x_min); atomicMax(&r->x_max, m.x_max);
                atomicMin(&r->y_min, m.y_min); atomicMax(&r->y_max, m.y_max);
                atomicAdd(&r->num_pixels, m.num_pixels);
                atomicAdd(&r->sum_i, m.sum_i);
                atomicAdd(&r->sum_xi, m.sum_xi);
                atomicAdd(&r->sum_yi, m.sum_yi);
                atomicMax(&r->peak, m.peak);
            }
        }
        __syncthreads();   // every find is done (the forest is not changed any more; also waits for this block's atomics to have been performed)
        FFS_PHASE_TS(A, 3);
        {
            uint32_t sl = block_exclusive_scan<kMergeThreads>(mine, s_wave, before);
            for (uint32_t i = i0; i < i1; ++i)
                if (s_par[i] == i) {   // (a root still points at itself: finds compress nothing)
                    if (sl < (uint32_t)kMergeCap) s_l2id[sl] = (uint16_t)i;
                    ++sl;
                }
        }
        __syncthreads();
        FFS_PHASE_TS(A, 4);
        // ---- records, kMergeThreads labels at a time, a thread per label: staged in LDS, out as consecutive dwords ------------------
        WireRec2* recs = reinterpret_cast<WireRec2*>(sa.recs) + (uint64_t)frame * A.rec_stride;
        const uint32_t ncomp = min(before, sa.max_comp);
        constexpr uint32_t kRound = (uint32_t)(kMergeLabels * kMergeThreads);
        for (uint32_t L0 = 0; L0 < ncomp; L0 += kRound) {
            unsigned long long v[kMergeLabels][7];   // sum I, sum (2x+1) I, sum (2y+1) I, peak | x_min, x_max | y_min, y_max | pixels, -
#pragma unroll
            for (int u = 0; u < kMergeLabels; ++u) {   // (both labels' loads are asked for before either record is computed)
                const uint32_t lab = L0 + (uint32_t)(u * kMergeThreads + tid);
                if (lab < ncomp) {
                    const uint32_t id = s_l2id[lab];
                    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(acc_of(id));
                    if ((s_touched[id >> 5] >> (id & 31u)) & 1u) {
                        // agent-scope loads: members were folded in by atomics at the L2
#pragma unroll
                        for (int w = 0; w < 7; ++w) v[u][w] = __hip_atomic_load(p + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
#pragma unroll
                        for (int w = 0; w < 7; ++w) v[u][w] = p[w];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kMergeLabels; ++u) {
                const uint32_t slot = (uint32_t)(u * kMergeThreads + tid);
                if (L0 + slot < ncomp) {
                    WireRec2 o;
                    chain_record(sa, W, (uint32_t)v[u][6], v[u][0], v[u][1], v[u][2], (uint32_t)v[u][4], (uint32_t)(v[u][4] >> 32), (uint32_t)v[u][5],
                                 (uint32_t)(v[u][5] >> 32), 0xFFFFFFFFu - (uint32_t)(v[u][3] & 0xFFFFFFFFull), (uint32_t)(v[u][3] >> 32), s_sm, o);
                    *reinterpret_cast<WireRec2*>(&s_out[slot * (sizeof(WireRec2) / 4)]) = o;
                }
            }
            __syncthreads();
            {
                const uint32_t here = min(kRound, ncomp - L0);
                uint32_t* dst = reinterpret_cast<uint32_t*>(recs + L0);
                const uint32_t ndw = here * (uint32_t)(sizeof(WireRec2) / 4);
                for (uint32_t w = tid; w < ndw; w += kMergeThreads) dst[w] = s_out[w];
            }
            __syncthreads();
        }
    }
    // ---- counters: as k_frame_chain ---------------------------------------------------------------------------------------------
    __syncthreads();
    FFS_PHASE_TS(A, 5);
    FFS_PHASE_TS(A, 6);
    if (tid == 0) {
        uint32_t fl = *a.overflow | flags;
        if (total > a.cap) fl |= 1u;
        if (before > sa.max_comp) fl |= 2u;
        a.num_strong[frame] = total;
        a.n_comp[frame] = before;
        const size_t Bn = A.max_batch;
        A.h_counts[frame] = total;
        A.h_counts[Bn + frame] = before;
        A.h_counts[10 * Bn + 1 + frame] = fl;
    }
    if (tid < 8) {
        a.summary[(uint64_t)frame * 8 + tid] = s_sm[tid];
        A.h_counts[2 * (size_t)A.max_batch + (size_t)frame * 8 + tid] = s_sm[tid];
    }
}

}  // namespace ffsamd
