// kernels_extended.hpp (included by ffs_submit.hip) -- the extended dispersion algorithm
// (`-a dispersion_extended`) for gfx950.
//
// What the reference does: DispersionExtendedThreshold::threshold, baseline/spotfinder/baseline.cpp:730-761
// (SAT -> dispersion mask -> erosion -> SAT over the eroded mask -> final threshold); on the device three
// dense byte-mask kernels, spotfinder/kernels/thresholding.cu:253-342, erosion.cu:53-143,
// thresholding.cu:360-491, each a shared-memory tile with a (2r+1)^2 loop per pixel.
//
// MI355X design: masks are bit planes (1 bit / pixel, as in the standard path), and only the first
// pass is dense.
//   X1 `k_ext_first`  streams the frame once: a wave64 marches down a 64-px column strip (one pixel per
//      lane, 56 owned) with a 7-row register ring; exact integer column sums {n, sum p, sum p^2}
//      slide vertically, the horizontal 7-tap comes from DPP wave shifts.  The oracle's fp64 test
//      a > c runs only on lanes that a float32 form with an explicit error allowance cannot reject.
//      Output: the "not background" plane D.
//   X2 `k_ext_erode`  5x5 binary erosion of D on the bit planes (shifts + ANDs, 32 px per lane):
//      the signal region E.  Pixels outside the image never erode; masked pixels erode
//      (baseline.cpp:552-571) or are skipped (erosion.cu:101-105) according to the flavour.
//   X3 `k_ext_final`  visits only the pixels of E (a few x the strong pixels): 11x11 sums of the
//      valid pixels outside E, then mean + nsig_s sqrt(mean) in fp64 (baseline.cpp:580-645).  Same
//      tile / compaction skeleton as the standard path's exact stage (exact_tile, MODE 1); it leaves
//      the strong plane, the byte mask and the per-tile counts for the connected-components stage.
#pragma once
#include "kernels_threshold.hpp"

namespace ffsamd {

constexpr int kExtOwnedPx = 56;    // lanes 4..59 of a wave own output
constexpr int kExtLaneOffset = 4;  // lane l sits on x = 56 * strip - 4 + l

template <typename PixelT>
__global__ __launch_bounds__(64) void k_ext_first(const ThresholdArgs a) {
    const int lane = threadIdx.x;
    const int strip = blockIdx.x % a.ext_strips;
    const int band = blockIdx.x / a.ext_strips;
    const int frame = blockIdx.y;
    const int yb0 = band * a.ext_band_rows;
    const int yb1 = min(yb0 + a.ext_band_rows, a.H);
    const int x = strip * kExtOwnedPx - kExtLaneOffset + lane;
    const bool in_x = x >= 0 && x < a.W;
    const uint8_t* img = (const uint8_t*)a.image + (uint64_t)frame * a.frame_stride;
    uint8_t* dplane = a.dplane + (uint64_t)frame * a.plane_frame_stride;

    uint32_t ring_p[7];   // masked pixel value of the last 7 rows
    uint32_t hist = 0;    // bit s: the pixel in ring slot s counted; bit 8+s: its mask bit
#pragma unroll
    for (int s = 0; s < 7; ++s) ring_p[s] = 0;
    int cm = 0;                    // column sums over the 7-row window
    uint32_t cx = 0;
    unsigned long long cy = 0;
    const int total = (yb1 - yb0) + 6;  // incoming rows yb0-3 .. yb1+2

    for (int base = 0;; base += 7) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int i = base + t;  // incoming row number; ring slot t
            if (i >= total) return;
            const int yin = yb0 - 3 + i;
            uint32_t p = 0, mbit = 0;
            if (in_x && yin >= 0 && yin < a.H) {
                p = reinterpret_cast<const PixelT*>(img + (uint64_t)yin * a.pitch)[x];
                mbit = (a.maskbits[(uint64_t)yin * a.mpitch + (x >> 3)] >> (x & 7)) & 1u;
            }
            // mm = mask && src < 2^24, baseline.cpp:379,391
            const bool inc = mbit && (sizeof(PixelT) == 2 || p < (1u << 24));
            const uint32_t pn = inc ? p : 0u, po = ring_p[t];
            cm += (int)inc - (int)((hist >> t) & 1u);
            cx += pn - po;
            cy += (unsigned long long)pn * pn - (unsigned long long)po * po;
            ring_p[t] = pn;
            hist = (hist & ~(0x101u << t)) | ((uint32_t)inc << t) | (mbit << (8 + t));
            if (i < 6) continue;  // window not full yet

            // horizontal 7-tap: neighbours at distance 1..3 on both sides by repeated wave shifts
            uint32_t wm = (uint32_t)cm, wx = cx;
            unsigned long long wy = cy;
            {
                uint32_t lm = (uint32_t)cm, lx = cx, ll = (uint32_t)cy, lh = (uint32_t)(cy >> 32);
                uint32_t rm = lm, rx = lx, rl = ll, rh = lh;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    lm = from_left(lm); lx = from_left(lx); ll = from_left(ll); lh = from_left(lh);
                    rm = from_right(rm); rx = from_right(rx); rl = from_right(rl); rh = from_right(rh);
                    wm += lm + rm;
                    wx += lx + rx;
                    wy += (((unsigned long long)lh << 32) | ll) + (((unsigned long long)rh << 32) | rl);
                }
            }
            // centre row = incoming row i - 3 = ring slot (t + 4) % 7
            constexpr int kDummy = 0;
            (void)kDummy;
            const int sc = (t + 4) % 7;
            const uint32_t centre_mask = (hist >> (8 + sc)) & 1u;
            const int m = (int)wm;
            bool d_bit = false;
            // baseline.cpp:469 (mask[k] && m >= min_count && x >= 0)
            if (centre_mask && m >= a.min_count) {
                // exact integers: a = m y - x^2 - x (m - 1)
                const long long ai = (long long)((unsigned long long)m * wy) - (long long)((unsigned long long)wx * wx)
                                     - (long long)((unsigned long long)wx * (unsigned)(m - 1));
                // float32 screen with allowance: all roundings together stay below 2^-21 relative
                const float af = (float)ai;
                const float cf = (float)wx * (a.kB * __builtin_sqrtf(2.0f * (float)(m - 1)));
                if (ai > 0 && af >= cf * (1.0f - 3.8146973e-06f)) {
                    // :470-472, each operation rounded separately (contraction is off)
                    const double md = (double)m, xd = (double)wx, yd = (double)wy;
                    const double t0 = md * yd;
                    const double t1 = xd * xd;
                    const double t2 = xd * (md - 1.0);
                    const double av = (t0 - t1) - t2;
                    const double cv = (xd * a.nsig_b) * __builtin_sqrt(2.0 * (md - 1.0));
                    d_bit = av > cv;
                }
                if (a.max_valid >= 0) {  // device kernels only, thresholding.cu:318-325
                    const uint32_t pc = reinterpret_cast<const PixelT*>(img + (uint64_t)(yin - 3) * a.pitch)[x];
                    if ((long long)pc > a.max_valid) d_bit = false;
                }
            }
            const bool owned = lane >= kExtLaneOffset && lane < kExtLaneOffset + kExtOwnedPx;
            const unsigned long long bal = __ballot(d_bit && owned) >> kExtLaneOffset;
            const uint32_t bcol = (uint32_t)strip * (kExtOwnedPx / 8) + (uint32_t)lane;
            if (lane < kExtOwnedPx / 8 && bcol < a.mpitch)
                dplane[(uint64_t)(yin - 3) * a.mpitch + bcol] = (uint8_t)(bal >> (8 * lane));
        }
    }
}
template __global__ void k_ext_first<uint16_t>(const ThresholdArgs);
template __global__ void k_ext_first<uint32_t>(const ThresholdArgs);

// One row of the horizontally eroded plane for word column w: row yy of D with "does not erode its neighbours" pixels set (beyond
// the image width; with the device kernels' rule also masked pixels), eroded horizontally by 2.  `centre` = D's own word.
__device__ __forceinline__ uint32_t ext_erode_hrow(const ThresholdArgs& a, const uint32_t* dp, const uint32_t* mp, int dpr, int w, int yy,
                                                   uint32_t beyond_c, uint32_t beyond_r, bool dev_rules, uint32_t& centre) {
    if (yy < 0 || yy >= a.H) { centre = 0; return ~0u; }
    const uint32_t* row = dp + (uint64_t)yy * dpr;
    const uint32_t* mrow = mp + (uint64_t)yy * dpr;
    centre = row[w];
    uint32_t c = centre | beyond_c, l = ~0u, r = ~0u;
    if (w > 0) l = row[w - 1];
    if (w + 1 < dpr) r = row[w + 1] | beyond_r;
    if (dev_rules) {
        c |= ~mrow[w];
        if (w > 0) l |= ~mrow[w - 1];
        if (w + 1 < dpr) r |= ~mrow[w + 1];
    }
    return c & ((c << 1) | (l >> 31)) & ((c << 2) | (l >> 30)) & ((c >> 1) | (r << 31)) & ((c >> 2) | (r << 30));
}

// X2: 5x5 erosion on bit planes.  One lane per 32-pixel word column and band of kErodeRows rows:
// the lane walks down its column with the horizontally eroded words of the last five rows in
// registers, so every plane word is fetched once (plus its two neighbours, which the adjacent
// lanes fetch as their own word: L1 hits).
constexpr int kErodeRows = 32;
__global__ __launch_bounds__(256) void k_ext_erode(const ThresholdArgs a) {
    const int dpr = (int)(a.mpitch >> 2);
    const int n_bands = (a.H + kErodeRows - 1) / kErodeRows;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int frame = blockIdx.y;
    if (t >= dpr * n_bands) return;
    const int w = t % dpr, band = t / dpr;
    const int y0 = band * kErodeRows, y1 = min(y0 + kErodeRows, a.H);
    // (__restrict__: the eroded plane is another buffer, so the loads of the next rows may pass the stores of this one --
    // with the loop unrolled eight times a lane has eight rows of loads in flight instead of one)
    const uint32_t* __restrict__ dp = reinterpret_cast<const uint32_t*>(a.dplane + (uint64_t)frame * a.plane_frame_stride);
    const uint32_t* __restrict__ mp = reinterpret_cast<const uint32_t*>(a.maskbits);
    uint32_t* __restrict__ ep = reinterpret_cast<uint32_t*>(a.eplane + (uint64_t)frame * a.plane_frame_stride);
    // bits of this word column that lie beyond the image width: they never erode anything
    const int x0 = w * 32;
    const uint32_t beyond_c = x0 + 32 > a.W ? (x0 >= a.W ? ~0u : ~((1u << (a.W - x0)) - 1u)) : 0u;
    const uint32_t beyond_r = x0 + 64 > a.W ? (x0 + 32 >= a.W ? ~0u : ~((1u << (a.W - x0 - 32)) - 1u)) : 0u;
    const bool dev_rules = a.ext_flavour == 1;  // masked pixels do not erode either (erosion.cu:101-105)
    auto hrow = [&](int yy, uint32_t& centre) -> uint32_t { return ext_erode_hrow(a, dp, mp, dpr, w, yy, beyond_c, beyond_r, dev_rules, centre); };
    uint32_t h0, h1, h2, h3, h4, c2, c3, c4, dummy;
    h0 = hrow(y0 - 2, dummy);
    h1 = hrow(y0 - 1, dummy);
    h2 = hrow(y0, c2);
    h3 = hrow(y0 + 1, c3);
#pragma unroll 8
    for (int y = y0; y < y1; ++y) {
        h4 = hrow(y + 2, c4);
        ep[(uint64_t)y * dpr + w] = c2 & h0 & h1 & h2 & h3 & h4;
        h0 = h1; h1 = h2; h2 = h3; h3 = h4;
        c2 = c3; c3 = c4;
    }
}

// X2, round 4: the same erosion with a wave per (62 word columns, band of ROWS rows).  k_ext_erode's lanes fetched three words
// per row (their own and both neighbours') eight rows at a time: a wave lived through five dependent round trips, 3-4 waves a
// CU on average (rocprofv3 counters, profiles/r04e_pmc_threshold_eiger16m_extended_b32.json: 0.18 GB in 54 us).  Here a lane
// fetches ONE word per row -- ROWS + 4 loads, all in flight before the first is used -- and takes its neighbours' words from
// the lanes beside it (DPP wave shifts; lanes 0 and 63 carry the words of the strips on either side and own no output).
// SPARSE: the signal-region plane is known to be zero (cleared behind the previous batch, like the first-pass plane), so only
// the words that hold a pixel of E are stored: a few per cent of them.
template <int ROWS, bool SPARSE>
__global__ __launch_bounds__(64) void k_ext_erode_strips(const ThresholdArgs a) {
    const int dpr = (int)(a.mpitch >> 2);
    const int n_strips = (dpr + 61) / 62;
    const int lane = threadIdx.x;
    // (XCD-aware block map, as the streaming kernels': the strips of a band share lines of the plane, so they share blockIdx % 8)
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int strip = qb % n_strips, band = xcd + 8 * (qb / n_strips);
    if (band * ROWS >= a.H) return;
    const int frame = blockIdx.y;
    const int w = strip * 62 + lane - 1;
    const bool in_w = w >= 0 && w < dpr;
    const int y0 = band * ROWS;
    const uint32_t* __restrict__ dp = reinterpret_cast<const uint32_t*>(a.dplane + (uint64_t)frame * a.plane_frame_stride);
    const uint32_t* __restrict__ mp = reinterpret_cast<const uint32_t*>(a.maskbits);
    uint32_t* __restrict__ ep = reinterpret_cast<uint32_t*>(a.eplane + (uint64_t)frame * a.plane_frame_stride);
    const rsrc_t r_d = make_rsrc(dp, (uint32_t)a.H * a.mpitch);
    const rsrc_t r_m = make_rsrc(mp, (uint32_t)a.H * a.mpitch);
    constexpr uint32_t kOob = 0x80000000u;
    const bool dev_rules = a.ext_flavour == 1;  // masked pixels do not erode either (erosion.cu:101-105)
    // bits of this word column that lie beyond the image width never erode anything; neither does a column outside the plane
    const int x0 = w * 32;
    const uint32_t beyond = !in_w ? ~0u : x0 + 32 > a.W ? (x0 >= a.W ? ~0u : ~((1u << (a.W - x0)) - 1u)) : 0u;
    uint32_t d[ROWS + 4], m[ROWS + 4];
#pragma unroll
    for (int r = 0; r < ROWS + 4; ++r) {
        const int yy = y0 - 2 + r;
        const uint32_t off = (in_w && yy >= 0 && yy < a.H) ? (uint32_t)w * 4u : kOob;   // (out of range reads 0)
        d[r] = __builtin_amdgcn_raw_buffer_load_b32(r_d, off, (uint32_t)max(yy, 0) * a.mpitch, 0);
        m[r] = ~0u;
    }
    if (dev_rules) {   // (uniform)
#pragma unroll
        for (int r = 0; r < ROWS + 4; ++r) {
            const int yy = y0 - 2 + r;
            const uint32_t off = (in_w && yy >= 0 && yy < a.H) ? (uint32_t)w * 4u : kOob;
            m[r] = __builtin_amdgcn_raw_buffer_load_b32(r_m, off, (uint32_t)max(yy, 0) * a.mpitch, 0);
        }
    }
    uint32_t h[ROWS + 4];
#pragma unroll
    for (int r = 0; r < ROWS + 4; ++r) {
        const int yy = y0 - 2 + r;
        uint32_t c = d[r] | beyond | ~m[r];
        if (yy < 0 || yy >= a.H) c = ~0u;        // rows outside the image never erode (uniform)
        const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)c, 0x138, 0xf, 0xf, false);   // wave_shr:1: lane i <- lane i - 1
        const uint32_t rr = (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)c, 0x130, 0xf, 0xf, false);  // wave_shl:1: lane i <- lane i + 1
        h[r] = c & ((c << 1) | (l >> 31)) & ((c << 2) | (l >> 30)) & ((c >> 1) | (rr << 31)) & ((c >> 2) | (rr << 30));
    }
    const bool owner = in_w && lane >= 1 && lane <= 62;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int y = y0 + r;
        if (y >= a.H) break;   // (uniform)
        const uint32_t e = d[r + 2] & h[r] & h[r + 1] & h[r + 2] & h[r + 3] & h[r + 4];
        if (owner && (!SPARSE || e != 0u)) ep[(uint64_t)y * dpr + w] = e;
    }
}
template __global__ void k_ext_erode_strips<32, false>(const ThresholdArgs);
template __global__ void k_ext_erode_strips<32, true>(const ThresholdArgs);
template __global__ void k_ext_erode_strips<16, false>(const ThresholdArgs);
template __global__ void k_ext_erode_strips<16, true>(const ThresholdArgs);

// X3 predicate: baseline.cpp:580-645 for one pixel of the signal region E.
template <typename PixelT>
__device__ __forceinline__ bool ext_final_strong(const ThresholdArgs& a, const uint8_t* img, const uint32_t* eplane, int e_y0, int x, int y) {
    const int W = a.W, H = a.H;
    const int xs = max(x - 5, 0), xe = min(x + 5, W - 1);  // kernel + 2, clipped (:591-598)
    const int ncol = xe - xs + 1;
    const int dpr = (int)(a.mpitch >> 2);
    const int w0 = xs >> 5, sh = xs & 31;
    const uint32_t colmask = (1u << ncol) - 1u;
    uint32_t m2 = 0;
    unsigned long long x2 = 0;
    if constexpr (sizeof(PixelT) == 2) {
        // 12 pixels from an even column cover the (<= 11 wide) window row: six aligned dword loads per row,
        // no data-dependent loop, four rows of loads in flight
        const int bx = min(xs & ~1, a.pitch_px - 12);
        const int wb = bx >> 5, shb = bx & 31;
        const uint32_t cm = (colmask << (xs - bx)) & 0xFFFu;
        const bool two = wb + 1 < dpr;
#pragma unroll 4
        for (int r = 0; r < 11; ++r) {
            const int yy = y - 5 + r;
            const bool ok = yy >= 0 && yy < H;
            const int yc = ok ? yy : y;
            const uint32_t* mrow = reinterpret_cast<const uint32_t*>(a.maskbits) + (uint64_t)yc * dpr;
            const uint32_t* erow = eplane + (int64_t)(yc - e_y0) * dpr;
            const unsigned long long mw = (unsigned long long)mrow[wb] | (two ? (unsigned long long)mrow[wb + 1] << 32 : 0ull);
            const unsigned long long ew = (unsigned long long)erow[wb] | (two ? (unsigned long long)erow[wb + 1] << 32 : 0ull);
            const uint32_t* prow = reinterpret_cast<const uint32_t*>(img + (uint64_t)yc * a.pitch + (uint64_t)bx * 2u);
            uint32_t pw[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) pw[q] = prow[q];
            // background for the second SAT: valid and not in the signal region (:552-571, :755)
            const uint32_t inc = ok ? ((uint32_t)((mw & ~ew) >> shb) & cm) : 0u;
            m2 += __popc(inc);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                x2 += ((inc >> (2 * q)) & 1u) ? (pw[q] & 0xFFFFu) : 0u;
                x2 += ((inc >> (2 * q + 1)) & 1u) ? (pw[q] >> 16) : 0u;
            }
        }
    } else {
        for (int yy = max(y - 5, 0); yy <= min(y + 5, H - 1); ++yy) {
            const uint32_t* mrow = reinterpret_cast<const uint32_t*>(a.maskbits) + (uint64_t)yy * dpr;
            const uint32_t* erow = eplane + (int64_t)(yy - e_y0) * dpr;
            const bool two = w0 + 1 < dpr;
            const unsigned long long mw = (unsigned long long)mrow[w0] | (two ? (unsigned long long)mrow[w0 + 1] << 32 : 0ull);
            const unsigned long long ew = (unsigned long long)erow[w0] | (two ? (unsigned long long)erow[w0 + 1] << 32 : 0ull);
            uint32_t inc = (uint32_t)((mw & ~ew) >> sh) & colmask;
            const PixelT* prow = reinterpret_cast<const PixelT*>(img + (uint64_t)yy * a.pitch) + xs;
            while (inc) {
                const int q = __ffs(inc) - 1;
                inc &= inc - 1;
                const uint32_t p = prow[q];
                if (p < (1u << 24)) {  // compute_sat's BIG, :379,391
                    m2 += 1;
                    x2 += p;
                }
            }
        }
    }
    const uint32_t pc = reinterpret_cast<const PixelT*>(img + (uint64_t)y * a.pitch)[x];
    if (a.ext_flavour == 1 && m2 == 0) return false;                     // thresholding.cu:472
    if (a.max_valid >= 0 && (long long)pc > a.max_valid) return false;  // thresholding.cu:440-441
    const double src = (double)pc;
    const double mean = m2 >= 2 ? (double)x2 / (double)m2 : 0.0;        // :640
    const bool global_mask = src > a.threshold;
    const bool local_mask = src >= (mean + a.nsig_s * __builtin_sqrt(mean));
    return global_mask && local_mask;
}

// X3 for FOUR neighbouring pixels of the signal region at once (16-bit pixels; x0 a multiple of 4, 8 <= x0, x0 + 12 <= pitch_px),
// by the FOUR lanes of a quad (sub = lane & 3, all four active).  The region is made of fat runs, and the 11 x 11 windows of
// a run's pixels overlap in 10 of their 11 columns: the quad loads the 11 rows of the 18 columns x0 - 8 .. x0 + 9 once
// (lane `sub` takes rows sub, sub + 4, sub + 8: four 8-byte loads and a dword per row, all in flight together, where a pixel
// on its own takes 11 x 6 dwords), each lane keeps the 14 column sums of the background pixels (valid and outside the
// region) and the four windows' counts over its rows, two DPP quad exchanges add them up, and lane `sub` slides the window to
// its pixel and takes the float64 test.  Same arithmetic as ext_final_strong (integer sums are order-independent).
// Columns beyond the image width hold no mask bits, rows outside it are skipped: the clipping of :591-598.
// Returns whether pixel x0 + sub is strong (callers ask only for pixels of the region).
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ bool ext_final_strong4(const ThresholdArgs& a, const uint8_t* img, const uint32_t* eplane, int e_y0, int x0, int y, int sub) {
    const int H = a.H;
    const int dpr = (int)(a.mpitch >> 2);
    const int bx = x0 - 8;                         // block column 0; a multiple of 4
    const int wb = bx >> 5, shb = bx & 31;         // (shb <= 28: the block's 20 bits lie inside two plane words)
    const bool two = wb + 1 < dpr;
    uint32_t c[14];                                // background sums of block columns 3 .. 16 = x0 - 5 .. x0 + 8
#pragma unroll
    for (int j = 0; j < 14; ++j) c[j] = 0;
    uint32_t m2[4] = {0, 0, 0, 0};
    uint32_t centre[2] = {0, 0};                   // the four centre pixels (block columns 8 .. 11 of row y)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int r = sub + 4 * t;                 // 0 .. 11; row 11 does not exist
        const int yy = y - 5 + r;
        const bool ok = r < 11 && yy >= 0 && yy < H;
        const int yc = ok ? yy : y;
        const uint32_t* mrow = reinterpret_cast<const uint32_t*>(a.maskbits) + (uint64_t)yc * dpr;
        const uint32_t* erow = eplane + (int64_t)(yc - e_y0) * dpr;
        const unsigned long long mw = (unsigned long long)mrow[wb] | (two ? (unsigned long long)mrow[wb + 1] << 32 : 0ull);
        const unsigned long long ew = (unsigned long long)erow[wb] | (two ? (unsigned long long)erow[wb + 1] << 32 : 0ull);
        const uint8_t* prow = img + (uint64_t)yc * a.pitch + (uint64_t)bx * 2u;   // 8-byte aligned
        uint32_t pw[9];                            // block columns 0 .. 17, two per word
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint2 v = *reinterpret_cast<const uint2*>(prow + 8 * q);
            pw[2 * q] = v.x;
            pw[2 * q + 1] = v.y;
        }
        pw[8] = *reinterpret_cast<const uint32_t*>(prow + 32);
        // background for the second SAT: valid and not in the signal region (:552-571, :755)
        const uint32_t inc = ok ? (uint32_t)((mw & ~ew) >> shb) & 0xFFFFFu : 0u;
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const int k = 3 + j;
            const uint32_t p = (k & 1) ? pw[k >> 1] >> 16 : pw[k >> 1] & 0xFFFFu;
            c[j] += ((inc >> k) & 1u) ? p : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) m2[i] += (uint32_t)__popc(inc & (0x7FFu << (3 + i)));
        if (r == 5) { centre[0] = pw[4]; centre[1] = pw[5]; }   // (lane 1, t = 1)
    }
#pragma unroll
    for (int j = 0; j < 14; ++j) c[j] = quad_sum(c[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) m2[i] = quad_sum(m2[i]);
    centre[0] = quad_sum(centre[0]);               // (three lanes add zeros)
    centre[1] = quad_sum(centre[1]);
    uint32_t xs[4];
    xs[0] = 0;
#pragma unroll
    for (int j = 0; j < 11; ++j) xs[0] += c[j];
#pragma unroll
    for (int i = 1; i < 4; ++i) xs[i] = xs[i - 1] - c[i - 1] + c[i + 10];
    const uint32_t x2 = sub == 0 ? xs[0] : sub == 1 ? xs[1] : sub == 2 ? xs[2] : xs[3];
    const uint32_t m = sub == 0 ? m2[0] : sub == 1 ? m2[1] : sub == 2 ? m2[2] : m2[3];
    const uint32_t cw = sub < 2 ? centre[0] : centre[1];
    const uint32_t pc = (sub & 1) ? cw >> 16 : cw & 0xFFFFu;
    if (a.ext_flavour == 1 && m == 0) return false;                     // thresholding.cu:472
    if (a.max_valid >= 0 && (long long)pc > a.max_valid) return false;  // thresholding.cu:440-441
    const double src = (double)pc;
    const double mean = m >= 2 ? (double)x2 / (double)m : 0.0;          // :640
    return src > a.threshold && src >= (mean + a.nsig_s * __builtin_sqrt(mean));
}

// (16-bit pixels: four pixels per quad of lanes, MODE 2 of the tile skeleton; 32-bit pixels: one pixel per lane)
// (seven workgroups a CU: 72 VGPRs without a vector spill where the compiler took 80 for six; 186-187 against 189-190 us with the erosion,
// profiles/r04x_ext_final_7_waves_per_simd_ab.txt -- eight need 64 VGPRs and spill: +40 us)
template <typename PixelT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_ext_final(const ThresholdArgs a) { exact_tile<PixelT, 256, kExactListCap, sizeof(PixelT) == 2 ? 2 : 1>(a); }
template __global__ void k_ext_final<uint16_t>(const ThresholdArgs);
template __global__ void k_ext_final<uint32_t>(const ThresholdArgs);
// erosion + final pass in one launch (16-bit pixels; dynamic LDS: 18 rows of the plane = 18 * mpitch bytes)
__global__ __launch_bounds__(256) void k_ext_erode_final(const ThresholdArgs a) { exact_tile<uint16_t, 256, kExactListCap, 3>(a); }

}  // namespace ffsamd
