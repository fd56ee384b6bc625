// kernels_threshold.hpp -- what the threshold kernels share (buffer resources, DPP lane shifts) and the EXACT stage:
// the oracle's predicate on window sums gathered from memory (`exact_strong`) and the tile kernel that applies a
// predicate to every pixel marked in a bit plane (`exact_tile`).
//
// What "bit-exact" is judged against: the float64 summed-area-table predicate of
// baseline/spotfinder/standalone.cc:113-174; the reference kernel it replaces: spotfinder/kernels/thresholding.cu:60-125.
// The hot path decides every pixel inside the streaming kernels of kernels_stream.hpp; what is here serves
//   * windows those kernels cannot vouch for (sum p >= 65536, 32-bit pixels >= 2^24): k_bright_fix gathers them, or --
//     when their list overflows, and as the A/B partner (`ffs_ctx_set_tuning("threshold_path", 1)`) -- they are marked in
//     the plane as candidates and k_exact filters the plane;
//   * the final pass of the extended algorithm (kernels_extended.hpp: k_ext_final = exact_tile over the eroded region).
// Round 1's conservative candidate kernels (k_candidates_*) are gone: the streaming kernels replaced them.
#pragma once
#include "ffs_device.h"

namespace ffsamd {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    // raw buffer, stride 0; DST_SEL/format word as in the CDNA guides (0x00020000)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

// lane i <- lane i-1 (lane 0 gets 0)
__device__ __forceinline__ uint32_t from_left(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
// lane i <- lane i+1 (lane 63 gets 0)
__device__ __forceinline__ uint32_t from_right(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}

constexpr int kQCap = 64;        // lane-group queue entries per wave of the streaming kernels (8 KB of LDS)

// ================================================================================================
// K2: exact predicate on candidates
// ================================================================================================

// The oracle's predicate on a window's exact sums (standalone.cc:165-170), shared by the gathered forms below.
template <bool DISP_ONLY>
__device__ __forceinline__ bool exact_decide(const ThresholdArgs& a, uint32_t m, unsigned long long sx, unsigned long long sy, uint32_t pc, bool centre_valid) {
    // :165  mask[k] && m >= min_count && x >= 0 && src[k] > threshold
    const double src = (double)pc;
    if (!(centre_valid && (int)m >= a.min_count && (DISP_ONLY || src > a.threshold))) return false;
    if (a.max_valid >= 0 && (long long)pc > a.max_valid) return false;  // GPU reference only, thresholding.cu:208-215
    const double md = (double)m, xd = (double)sx, yd = (double)sy;
    // :166-170, each operation rounded separately (contraction is off for this library)
    const double t0 = md * yd;
    const double t1 = xd * xd;
    const double t2 = xd * (md - 1.0);
    const double av = (t0 - t1) - t2;
    const double bv = md * src - xd;
    const double cv = (xd * a.nsig_b) * __builtin_sqrt(2.0 * (md - 1.0));
    const double dv = a.nsig_s * __builtin_sqrt(xd * md);
    if constexpr (DISP_ONLY) return av > cv;
    return av > cv && bv > dv;
}

// Exact integer window sums + the oracle predicate, standalone.cc:113-174 operation for operation.
// All seven window rows (pixels and mask bits) are requested before any is used, so a candidate
// costs one memory round trip, not seven.
// DISP_ONLY: the extended algorithm's first pass (baseline.cpp:468-473) -- same sums, a > c alone.
template <typename PixelT, bool DISP_ONLY = false>
__device__ bool exact_strong(const ThresholdArgs& a, const uint8_t* img, int x, int y) {
    const int W = a.W, H = a.H;
    const int xs = max(x - 3, 0), xe = min(x + 3, W - 1);  // window clipped to the image, :126-130
    // 8 pixels starting at an even column cover the (<= 7 wide) window row
    const int bx = min(xs & ~1, a.pitch_px - 8);
    const uint32_t rm = ((1u << (xe - bx + 1)) - 1u) & ~((1u << (xs - bx)) - 1u);
    const int sh = bx & 7;

    uint4 r0[7], r1[7];
    uint32_t mb[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int yy = y - 3 + r;
        const bool ok = yy >= 0 && yy < H;  // rows outside the image contribute nothing
        const int yc = ok ? yy : y;
        const uint8_t* mp = a.maskbits + (uint64_t)yc * a.mpitch + (bx >> 3);
        uint32_t b = mp[0];
        if (sh) b |= (uint32_t)mp[1] << 8;
        mb[r] = ok ? b : 0u;
        const uint8_t* rp = img + (uint64_t)yc * a.pitch + (uint64_t)bx * sizeof(PixelT);
        r0[r] = *reinterpret_cast<const uint4*>(rp);  // 4-byte aligned
        if constexpr (sizeof(PixelT) == 4) r1[r] = *reinterpret_cast<const uint4*>(rp + 16);
    }

    uint32_t m = 0;
    unsigned long long sx = 0, sy = 0;
    uint32_t pc = 0;
    bool centre_valid = false;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        uint32_t p[8];
        if constexpr (sizeof(PixelT) == 2) {
            p[0] = r0[r].x & 0xFFFFu; p[1] = r0[r].x >> 16; p[2] = r0[r].y & 0xFFFFu; p[3] = r0[r].y >> 16;
            p[4] = r0[r].z & 0xFFFFu; p[5] = r0[r].z >> 16; p[6] = r0[r].w & 0xFFFFu; p[7] = r0[r].w >> 16;
        } else {
            p[0] = r0[r].x; p[1] = r0[r].y; p[2] = r0[r].z; p[3] = r0[r].w;
            p[4] = r1[r].x; p[5] = r1[r].y; p[6] = r1[r].z; p[7] = r1[r].w;
        }
        const uint32_t bits = (mb[r] >> sh) & rm;
        if (r == 3) {  // the candidate itself
            const int q = x - bx;
            centre_valid = (mb[r] >> (sh + q)) & 1u;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j == q) pc = p[j];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // mm = mask && src < 2^24, standalone.cc:78,90
            const bool inc = ((bits >> q) & 1u) && (sizeof(PixelT) == 2 || p[q] < (1u << 24));
            const uint32_t pv = inc ? p[q] : 0u;
            m += inc ? 1u : 0u;
            sx += pv;
            sy += (unsigned long long)pv * pv;
        }
    }

    return exact_decide<DISP_ONLY>(a, m, sx, sy, pc, centre_valid);
}

// The same decision from the same sums with HALF the registers: the window's rows come in two batches (four, then three) -- two
// memory round trips instead of one, 16 + 4 registers of pixels and mask bits in flight instead of 28 + 7.  For callers that must
// stay small (kernels_band.hpp: a one-wave workgroup that has to fit where a streaming wave has left).
template <typename PixelT>
__device__ __forceinline__ bool exact_strong_lite(const ThresholdArgs& a, const uint8_t* img, int x, int y) {
    const int W = a.W, H = a.H;
    const int xs = max(x - 3, 0), xe = min(x + 3, W - 1);
    const int bx = min(xs & ~1, a.pitch_px - 8);
    const uint32_t rm = ((1u << (xe - bx + 1)) - 1u) & ~((1u << (xs - bx)) - 1u);
    const int sh = bx & 7;
    uint32_t m = 0, pc = 0;
    unsigned long long sx = 0, sy = 0;
    bool centre_valid = false;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int r_lo = half * 4, nr = half ? 3 : 4;
        uint4 r0[4], r1[4];
        uint32_t mb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y - 3 + r_lo + q;
            const bool ok = q < nr && yy >= 0 && yy < H;
            const int yc = ok ? yy : y;
            const uint8_t* mp = a.maskbits + (uint64_t)yc * a.mpitch + (bx >> 3);
            uint32_t b = mp[0];
            if (sh) b |= (uint32_t)mp[1] << 8;
            mb[q] = ok ? b : 0u;
            const uint8_t* rp = img + (uint64_t)yc * a.pitch + (uint64_t)bx * sizeof(PixelT);
            r0[q] = *reinterpret_cast<const uint4*>(rp);
            if constexpr (sizeof(PixelT) == 4) r1[q] = *reinterpret_cast<const uint4*>(rp + 16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t p[8];
            if constexpr (sizeof(PixelT) == 2) {
                p[0] = r0[q].x & 0xFFFFu; p[1] = r0[q].x >> 16; p[2] = r0[q].y & 0xFFFFu; p[3] = r0[q].y >> 16;
                p[4] = r0[q].z & 0xFFFFu; p[5] = r0[q].z >> 16; p[6] = r0[q].w & 0xFFFFu; p[7] = r0[q].w >> 16;
            } else {
                p[0] = r0[q].x; p[1] = r0[q].y; p[2] = r0[q].z; p[3] = r0[q].w;
                p[4] = r1[q].x; p[5] = r1[q].y; p[6] = r1[q].z; p[7] = r1[q].w;
            }
            const uint32_t bits = (mb[q] >> sh) & rm;
            if (r_lo + q == 3) {  // the candidate itself
                const int c = x - bx;
                centre_valid = (mb[q] >> (sh + c)) & 1u;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j == c) pc = p[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool inc = ((bits >> j) & 1u) && (sizeof(PixelT) == 2 || p[j] < (1u << 24));
                const uint32_t pv = inc ? p[j] : 0u;
                m += inc ? 1u : 0u;
                sx += pv;
                sy += (unsigned long long)pv * pv;
            }
        }
    }
    return exact_decide<false>(a, m, sx, sy, pc, centre_valid);
}

// extended algorithm's final test (kernels_extended.hpp)
// (the signal-region plane E: `eplane` = the frame's plane in global memory, row 0 first, rows `edpr` dwords apart; MODE 3 passes
// the tile's rows of it in LDS instead -- row `e_y0` first)
template <typename PixelT>
__device__ __forceinline__ bool ext_final_strong(const ThresholdArgs& a, const uint8_t* img, const uint32_t* eplane, int e_y0, int x, int y);
__device__ __forceinline__ bool ext_final_strong4(const ThresholdArgs& a, const uint8_t* img, const uint32_t* eplane, int e_y0, int x0, int y, int sub);
// 5 x 5 erosion of the first-pass plane D, one row of one 32-pixel word column: see kernels_extended.hpp
__device__ __forceinline__ uint32_t ext_erode_hrow(const ThresholdArgs& a, const uint32_t* dp, const uint32_t* mp, int dpr, int w, int yy,
                                                   uint32_t beyond_c, uint32_t beyond_r, bool dev_rules, uint32_t& centre);

// MODE 0: candidates come from (and strong pixels go back to) a.bits, predicate exact_strong.
// MODE 1: extended algorithm -- candidates are the signal-region plane a.eplane (read-only: other
//         tiles read it for their 11x11 windows), predicate ext_final_strong, result in a.bits.
// MODE 2: the same with a list entry per aligned group of FOUR pixels that holds a candidate (16-bit pixels,
//         ext_final_strong4: the four windows share their loads and column sums).
// MODE 3: MODE 2 with the erosion fused in (round 4): the tile computes the rows y0 - 5 .. y0 + 12 of the signal region E from rows
//         y0 - 7 .. y0 + 14 of the first-pass plane a.dplane into LDS (dynamic: 18 rows of the plane), writes its own eight rows
//         of E to a.eplane (for --writeout / ffs_stream_debug_bitplane) and takes every window's E bits from LDS: one launch and a
//         round trip of the plane less than k_ext_erode + MODE 2.
template <typename PixelT, int NT, int LISTCAP, int MODE = 0>
__device__ __forceinline__ void exact_tile(const ThresholdArgs& a) {
    // The stage is latency-bound (sparse gathers).  Measured dead ends: a smaller LDS footprint
    // (more tiles resident) and one-wave workgroups both made it slower.  Round 4, extended algorithm (MODE 2, profiles/r04r_ext_final_*):
    // of 155 us per 32 Eiger frames the tiles' heads (their words of the plane) are 14, the candidates' windows 92, lists / barriers /
    // results 50.  Tried and dropped, each bit-exact: a third fewer vector instructions per candidate (v_dot2 sums, a float32 test
    // ahead of the float64 one), the head's loads all in flight, an XCD-aware tile map -- no difference or 3 % slower together; 64 VGPRs
    // for eight tiles a CU -- spills, +40 us; a workgroup per RANGE of tiles with the next tile's words prefetched -- 109 VGPRs, four
    // workgroups a CU, +55 us; 16-byte pixel gathers and 8-byte plane gathers (five lane-loads a row instead of nine) -- +100 us.
    __shared__ uint32_t s_words[kTileRows * 320];  // tile bit-plane words (pitch_px <= 10240)
    __shared__ uint32_t s_list[LISTCAP];
    __shared__ uint32_t s_cnt, s_total, s_strong;

    const int tid = threadIdx.x;
    const int tile = blockIdx.x, frame = blockIdx.y;
    const int y0 = tile * kTileRows;
    const int rows = min(kTileRows, a.H - y0);
    const int dpr = a.mpitch >> 2;  // dwords per row
    const int ndw = rows * dpr;
    const uint8_t* img = (const uint8_t*)a.image + (uint64_t)frame * a.frame_stride;
    uint32_t* gwords = reinterpret_cast<uint32_t*>(a.bits + (uint64_t)frame * a.plane_frame_stride
                                                   + (uint64_t)y0 * a.mpitch);
    const uint8_t* eframe = MODE >= 1 ? a.eplane + (uint64_t)frame * a.plane_frame_stride : nullptr;
    const uint32_t* gin = MODE >= 1 ? reinterpret_cast<const uint32_t*>(eframe + (uint64_t)y0 * a.mpitch) : gwords;
    uint8_t* sbytes = a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride;
    // where the windows' E bits come from: the frame's plane in global memory (row 0 first), or the tile's 18 rows in LDS
    [[maybe_unused]] const uint32_t* esrc = reinterpret_cast<const uint32_t*>(eframe);
    [[maybe_unused]] int e_y0 = 0;

    if (tid == 0) { s_cnt = 0; s_total = 0; s_strong = 0; }
    if constexpr (MODE == 3) {
        extern __shared__ uint32_t s_E[];   // [18][dpr]: rows y0 - 5 .. y0 + 12 of the signal region
        constexpr int kERows = kTileRows + 10;
        const uint32_t* dp = reinterpret_cast<const uint32_t*>(a.dplane + (uint64_t)frame * a.plane_frame_stride);
        const uint32_t* mp = reinterpret_cast<const uint32_t*>(a.maskbits);
        const bool dev_rules = a.ext_flavour == 1;
        // two halves of the rows per word column, each marching down with the horizontally eroded words of five rows in registers
        for (int item = tid; item < 2 * dpr; item += NT) {
            const int half = item >= dpr ? 1 : 0, w = item - half * dpr;
            const int x0 = w * 32;
            const uint32_t beyond_c = x0 + 32 > a.W ? (x0 >= a.W ? ~0u : ~((1u << (a.W - x0)) - 1u)) : 0u;
            const uint32_t beyond_r = x0 + 64 > a.W ? (x0 + 32 >= a.W ? ~0u : ~((1u << (a.W - x0 - 32)) - 1u)) : 0u;
            const int r0 = half * (kERows / 2), r1 = r0 + kERows / 2;          // E rows [r0, r1) of the 18
            const int ya = y0 - 5 + r0;                                          // first image row of this half
            uint32_t h0, h1, h2, h3, h4, c2, c3, c4, dummy;
            h0 = ext_erode_hrow(a, dp, mp, dpr, w, ya - 2, beyond_c, beyond_r, dev_rules, dummy);
            h1 = ext_erode_hrow(a, dp, mp, dpr, w, ya - 1, beyond_c, beyond_r, dev_rules, dummy);
            h2 = ext_erode_hrow(a, dp, mp, dpr, w, ya, beyond_c, beyond_r, dev_rules, c2);
            h3 = ext_erode_hrow(a, dp, mp, dpr, w, ya + 1, beyond_c, beyond_r, dev_rules, c3);
#pragma unroll
            for (int r = 0; r < kERows / 2; ++r) {
                h4 = ext_erode_hrow(a, dp, mp, dpr, w, ya + r + 2, beyond_c, beyond_r, dev_rules, c4);
                s_E[(r0 + r) * dpr + w] = c2 & h0 & h1 & h2 & h3 & h4;   // (rows outside the image: c2 = 0)
                h0 = h1; h1 = h2; h2 = h3; h3 = h4;
                c2 = c3; c3 = c4;
            }
            (void)r1;
        }
        __syncthreads();
        // the tile's own rows of E leave for the global plane (what k_ext_erode wrote)
        uint32_t* eout = reinterpret_cast<uint32_t*>(a.eplane + (uint64_t)frame * a.plane_frame_stride + (uint64_t)y0 * a.mpitch);
        for (int g = tid; g < ndw; g += NT) eout[g] = s_E[5 * dpr + g];
        esrc = s_E;
        e_y0 = y0 - 5;
    }
    // what a list entry stands for: a candidate pixel, or (MODE 2) the first bit of a group of four that holds one
    auto entries = [](uint32_t w) -> uint32_t { return MODE >= 2 ? (w | (w >> 1) | (w >> 2) | (w >> 3)) & 0x11111111u : w; };
    uint32_t mine = 0;
    for (int g = tid; g < ndw; g += NT) {
        uint32_t w;
        if constexpr (MODE == 3) w = esrc[5 * dpr + g];
        else w = gin[g];
        s_words[g] = w;
        mine += __popc(entries(w));
    }
    __syncthreads();
    if (FFS_DBG(a, 2048)) return;   // (experiments build: what the tiles' head costs)
    if (mine) atomicAdd(&s_total, mine);
    __syncthreads();
    const uint32_t total = s_total;  // block-uniform
    if (total == 0) {
        if (tid == 0) a.tile_counts[(uint64_t)frame * a.n_tiles + tile] = 0;
        if constexpr (MODE >= 1)
            for (int g = tid; g < ndw; g += NT) gwords[g] = 0;
        return;
    }

    auto append = [&](int g, uint32_t w, uint32_t at) {
        while (w) {
            const uint32_t bit = __ffs(w) - 1;
            w &= w - 1;
            s_list[at++] = ((uint32_t)g << 5) | bit;
        }
    };
    auto flush = [&]() {
        const uint32_t n = FFS_DBG(a, 512) ? 0u : s_cnt;   // (experiments build: no candidate is looked at)
        if constexpr (MODE >= 2) {
            // an entry = an aligned group of four pixels, taken by a quad of lanes (whole quads are in or out of the loop)
            const int sub = tid & 3;
            for (uint32_t e = (uint32_t)tid >> 2; e < n; e += NT / 4) {
                const uint32_t idx = s_list[e];
                const uint32_t g = idx >> 5, bit = idx & 31u;
                const int row = g / dpr;
                const int x = (int)((g - row * dpr) * 32u + bit);
                const int y = y0 + row;
                const bool want = (s_words[g] >> (bit + (uint32_t)sub)) & 1u;   // (only this lane ever changes this bit)
                bool strong;
                if (x >= 8 && x + 12 <= a.pitch_px) strong = ext_final_strong4(a, img, esrc, e_y0, x, y, sub);   // (quad-uniform branch)
                else strong = want && ext_final_strong<PixelT>(a, img, esrc, e_y0, x + sub, y);                   // next to the frame's left or right edge
                if (want) {
                    if (strong) sbytes[(uint64_t)y * a.bpitch + x + sub] = 1;
                    else atomicAnd(&s_words[g], ~(1u << (bit + (uint32_t)sub)));
                }
            }
            return;
        }
        for (uint32_t e = tid; e < n; e += NT) {
            const uint32_t idx = s_list[e];
            const uint32_t g = idx >> 5, bit = idx & 31u;
            const int row = g / dpr;
            const int x = (int)((g - row * dpr) * 32u + bit);
            const int y = y0 + row;
            bool strong;
            if constexpr (MODE == 1) strong = ext_final_strong<PixelT>(a, img, esrc, e_y0, x, y);
            else strong = exact_strong<PixelT>(a, img, x, y);
            if (strong) {
                sbytes[(uint64_t)y * a.bpitch + x] = 1;
            } else {
                atomicAnd(&s_words[g], ~(1u << bit));
            }
        }
    };

    if (total <= (uint32_t)LISTCAP) {
        // the usual case: every candidate of the tile in one dense pass
        if (mine) {
            uint32_t at = atomicAdd(&s_cnt, mine);
            for (int g = tid; g < ndw; g += NT) {
                const uint32_t w = entries(s_words[g]);
                append(g, w, at);
                at += __popc(w);
            }
        }
        __syncthreads();
        flush();
    } else {
        // dense tile: 64 words (<= 2048 candidates) at a time
        for (int pos = 0; pos < ndw; pos += LISTCAP / 32) {
            const int g = pos + tid;
            const uint32_t w = (tid < LISTCAP / 32 && g < ndw) ? entries(s_words[g]) : 0u;
            if (w) append(g, w, atomicAdd(&s_cnt, (uint32_t)__popc(w)));
            __syncthreads();
            flush();
            __syncthreads();
            if (tid == 0) s_cnt = 0;
            __syncthreads();
        }
    }
    __syncthreads();

    uint32_t cnt = 0;
    for (int g = tid; g < ndw; g += NT) {
        const uint32_t w = s_words[g];
        if (!FFS_DBG(a, 1024) || w) gwords[g] = w;   // (experiments build, bit 1024: the non-zero words only)
        cnt += __popc(w);
        if constexpr (MODE >= 1) {
            // the extended algorithm's final plane: its occupancy bitmap (one bit per 16-byte segment of a plane row) lets the
            // sparse stage read only the segments that hold something, as after the streaming kernels
            if (w) {
                const int row = g / dpr;
                const uint32_t ob = (uint32_t)(y0 + row) * a.occ_spr + (uint32_t)((g - row * dpr) >> 2);
                atomicOr(a.occ + (uint64_t)frame * a.occ_frame_words + (ob >> 5), 1u << (ob & 31u));
            }
        }
    }
    if (cnt) atomicAdd(&s_strong, cnt);
    __syncthreads();
    if (tid == 0) a.tile_counts[(uint64_t)frame * a.n_tiles + tile] = s_strong;
}

// NB: __launch_bounds__ must be a literal here -- with a template parameter hipcc 7.2 silently
// dropped it (default 1024-thread bound -> 178 VGPRs + scratch, kernel 2x slower).
template <typename PixelT>
__global__ __launch_bounds__(256) void k_exact(const ThresholdArgs a) { exact_tile<PixelT, 256, kExactListCap>(a); }
template __global__ void k_exact<uint16_t>(const ThresholdArgs);
template __global__ void k_exact<uint32_t>(const ThresholdArgs);
// Inclusive prefix sum over the 64 lanes in six DPP adds (row_shr 1/2/4/8 inside the rows of 16, then row_bcast:15 and
// row_bcast:31 carry the row totals on) instead of six ds_bpermute round trips.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}

// Inclusive running maximum over the 64 lanes, the same six DPP steps.
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

}  // namespace ffsamd
