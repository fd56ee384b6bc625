// kernels_stream.hpp (included by ffs_submit.hip) -- the whole dispersion threshold in ONE streaming kernel
// for 16-bit pixels on gfx950 (CDNA4): `k_stream_u16`.
//
// What the reference does: one 7x7 masked window sum per pixel from a shared-memory tile and a float32
// test (spotfinder/kernels/thresholding.cu:60-125, :145-234).  What "bit-exact" is judged against: the
// float64 summed-area-table predicate of baseline/spotfinder/standalone.cc:113-174.
//
// Design (round 2; replaces k_candidates_u16<true> + k_exact_w64 on the standard path):
//  * The valid-pixel mask never changes between frames, so everything that depends on it alone is a
//    table built once per mask (k_build_maps): the 7x7 window count m of every pixel (`mmap`, 1 B/px,
//    read only for the few queued groups) and, per 8-pixel lane group, one dword `ginfo` = the group's
//    8 mask bits | smallest m << 8 | largest m << 16 over its valid pixels.  The streaming loop then
//    carries pure pixel sums: no count field to drag through every add, no per-row mask expansion
//    (the four pair masks of a lane change only where the mask byte differs from the row above).
//  * Cost model (tools/ubench/valu_rate.hip, measured on MI355X): VOP2 add/sub/and/or/shift and
//    v_mul_f32/v_add_f32 issue at ~1.0 ns per wave per SIMD, everything else (VOP3, SDWA, DPP, packed,
//    24-bit multiplies, min/max, conversions, compares) at ~1.9 ns.  The row loop is written against
//    that: pixels are unpacked with one AND / one shift, the vertical sums are add/sub plus ONE
//    multiply-add per pixel for the sum of squares, the horizontal 7-tap is built from pair sums so
//    that every cross-lane operand rides on an add (v_add_u32_dpp) instead of a separate move.
//  * All frames of a batch lie side by side in one "super row": lane groups are numbered across
//    frames with one empty separator group between frames (it supplies the zeros outside the image),
//    so only the last strip of the whole batch is partly filled -- not the last strip of every frame
//    (Eiger-16M: 8.4 instead of 9 waves per row and frame).
//  * Group screen per lane and row: largest centre pixel against smallest window sum, with the group's
//    smallest / largest count from `ginfo` -- a proven superset of the oracle's signal test.  The few
//    groups that pass (a few %) are queued in LDS with their window sums, centre pixels and the 14
//    column sums of p^2; 64 queued groups at a time take, on dense lanes, the conservative float32
//    signal + dispersion tests and then, for the pixels still standing, the oracle's float64 predicate
//    itself, operation for operation (standalone.cc:165-170) on exact integer sums.  The plane
//    written is therefore the final strong-pixel plane: no second kernel, no re-read of the frame.
//    (Windows with sum p >= 65536 -- sum p^2 may exceed 32 bits -- fall back to exact_strong's gather.)
//  * Output: the non-zero bytes of the strong bit plane (the plane is all zero when the kernel starts: the
//    compaction kernel clears every word it has consumed), the zero-filled byte mask (whole 128-byte lines,
//    non-temporal; the 1s are set by the compaction kernel from the list) and per-tile strong counts
//    (atomics, about one per strong group).  Measured (tools/ubench/hbm_pattern.hip): byte-wise zero stores
//    into the plane cost 45-55 us per 32 frames on their own, and the memory system moves this 2:1 read/write
//    mix at 5.0 TB/s at best (352 us per 32 Eiger frames with no arithmetic at all).
#pragma once
#include "kernels_threshold.hpp"

// cache policy of the pixel loads of the streaming kernels (raw buffer load `aux`: 0 = default, 2 = non-temporal)
#ifndef FFS_IMG_LOAD_AUX
#define FFS_IMG_LOAD_AUX 0
#endif

namespace ffsamd {

constexpr int kSQWords = 32;         // queue entry: 0-7 window sums, 8-11 centre pixels (two per word), 12-13 window counts,
                                     // 14 result bits, 15 ginfo, 16-29 column sums of p^2, 30 tag (row << 6 | lane), 31 frame in super row << 16 | group

// The oracle's predicate on exact integer window sums, standalone.cc:165-170 operation for operation
// (the same lines as exact_strong, which gets its sums by gathering the window from memory).
__device__ __forceinline__ bool exact_predicate(const ThresholdArgs& a, uint32_t m, unsigned long long sx,
                                                unsigned long long sy, uint32_t pc) {
    const double src = (double)pc;
    if (!((int)m >= a.min_count && src > a.threshold)) return false;  // (an invalid centre has pc = 0 here)
    if (a.max_valid >= 0 && (long long)pc > a.max_valid) return false;  // GPU reference only, thresholding.cu:208-215
    const double md = (double)m, xd = (double)sx, yd = (double)sy;
    const double t0 = md * yd;
    const double t1 = xd * xd;
    const double t2 = xd * (md - 1.0);
    const double av = (t0 - t1) - t2;
    const double bv = md * src - xd;
    const double cv = (xd * a.nsig_b) * __builtin_sqrt(2.0 * (md - 1.0));
    const double dv = a.nsig_s * __builtin_sqrt(xd * md);
    return av > cv && bv > dv;
}

// The same decision without the two square roots, for m y < 2^53 (then a = m y - x^2 - x (m - 1) and
// b = m p - x are exact integers in float64, as they are in the oracle; the callers check the range).  The oracle compares a with
// c = fl(fl(x nsig_b) fl(sqrt(2 (m-1)))) and b with d = fl(nsig_s fl(sqrt(x m))): each within 2^-51 of
// the real number.  Comparing the squares instead, a^2 against nsig_b^2 x^2 2 (m-1) and b^2 against
// nsig_s^2 x m (a few float64 roundings each), settles the comparison whenever the two sides differ by
// more than 2^-40 relative -- which is always, unless they are equal as real numbers; `certain` says so,
// and the caller falls back to exact_predicate (with its correctly rounded square roots) otherwise.
__device__ __forceinline__ bool exact_predicate_nosqrt(const ThresholdArgs& a, uint32_t m, unsigned long long sx,
                                                       unsigned long long sy, uint32_t pc, bool& certain) {
    // (no early return, as in int_predicate: a wave runs all of it as long as one lane goes on)
    const double src = (double)pc;
    const bool ok = (int)m >= a.min_count && src > a.threshold && !(a.max_valid >= 0 && (long long)pc > a.max_valid);
    const double md = (double)m, xd = (double)sx, yd = (double)sy;
    const double av = (md * yd - xd * xd) - xd * (md - 1.0);
    const double bv = md * src - xd;
    const bool pos = av > 0.0 && bv > 0.0;  // (else not strong: c >= 0 and d >= 0)
    const double a2 = av * av, c2 = (a.nsig_b2 * (xd * xd)) * (2.0 * (md - 1.0));
    const double b2 = bv * bv, d2 = a.nsig_s2 * (xd * md);
    constexpr double kEps = 9.094947017729282e-13;  // 2^-40
    const bool disp_yes = a2 > c2 + c2 * kEps, disp_no = a2 < c2 - c2 * kEps;
    const bool sig_yes = b2 > d2 + d2 * kEps, sig_no = b2 < d2 - d2 * kEps;
    certain = !(ok && pos) || ((disp_yes || disp_no) && (sig_yes || sig_no));
    return ok && pos && disp_yes && sig_yes;
}

// The same decision in integers, for integer nsig_b, nsig_s (the defaults 6 and 3) and a window whose sums fit: x < 65536, so
// y < 2^32 and everything below stays inside 64 bits.  a = m y - x (x + m - 1) and b = m p - x are the oracle's a and b exactly;
// c^2 = nsig_b^2 x^2 2 (m - 1) and d^2 = nsig_s^2 x m are integers here, so the comparisons of the squares are exact, and
// whenever they differ by at least 16 the sides themselves differ by more than 2^-46 relative -- far beyond what the oracle's
// three roundings of c and d (2^-51) can turn around.  Closer than that (ties: 2 (m - 1) or x m a perfect square): `certain`
// is false and the caller evaluates exact_predicate.  A fifth of the float64 form's issue slots (v_mad_u64_u32 instead of
// chains of v_mul_f64 / v_fma_f64 and conversions).
__device__ __forceinline__ bool int_predicate(const ThresholdArgs& a, uint32_t m, uint32_t x, uint32_t y, uint32_t pc, bool& certain) {
    // (no early return: every `if (...) return false` was an exec-mask region of its own in each of the drain's ten inlined copies,
    // and a wave runs all of it anyway as long as one lane goes on -- DESIGN.md section 3.2e)
    const bool ok = (int)m >= a.min_count && pc > a.thr_floor && !(a.max_valid >= 0 && (long long)pc > a.max_valid);
    const unsigned long long my = (unsigned long long)m * y;                 // < 2^38
    const unsigned long long xx = (unsigned long long)x * (x + m - 1u);      // < 2^33
    const int32_t bv = (int32_t)__umul24(m, pc) - (int32_t)x;                // |b| < 2^22
    const bool pos = my > xx && bv > 0;                                      // else a <= 0 or b <= 0, and c, d >= 0: not strong
    const unsigned long long av = my - xx;                                   // (wraps where !pos: unused there)
    const unsigned long long c2 = (unsigned long long)(a.ib2 * 2u * (m - 1u)) * (unsigned long long)(x * x);   // < 2^17 2^32
    const bool small = av < (1ull << 25);                                    // (beyond: a^2 >= 2^50 > c^2)
    const unsigned long long a2 = (unsigned long long)(uint32_t)av * (uint32_t)av;
    const bool disp_yes = !small || a2 > c2;
    const bool disp_close = small && (a2 > c2 ? a2 - c2 : c2 - a2) < 16ull;
    const uint32_t bu = (uint32_t)bv;
    const unsigned long long b2 = (unsigned long long)bu * bu;               // < 2^44 where pos
    const unsigned long long d2 = (unsigned long long)a.is2 * __umul24(x, m);       // x m < 2^22
    const bool sig_yes = b2 > d2;
    const bool sig_close = (b2 > d2 ? b2 - d2 : d2 - b2) < 16ull;
    certain = !(ok && pos && (disp_close || sig_close));
    return ok && pos && disp_yes && sig_yes;
}

// First pass of the extended algorithm (baseline.cpp:468-473): the dispersion half alone, a > c, for a valid
// centre with at least min_count pixels in its window (and, with the device kernels' rule, a centre pixel not
// above max_valid).  Square-root-free with the same certification as above; dispersion_only() is the fall-back.
__device__ __forceinline__ bool dispersion_only(const ThresholdArgs& a, uint32_t m, unsigned long long sx, unsigned long long sy) {
    const double md = (double)m, xd = (double)sx, yd = (double)sy;
    const double t0 = md * yd;
    const double t1 = xd * xd;
    const double t2 = xd * (md - 1.0);
    const double av = (t0 - t1) - t2;
    const double cv = (xd * a.nsig_b) * __builtin_sqrt(2.0 * (md - 1.0));
    return av > cv;
}
__device__ __forceinline__ bool dispersion_only_nosqrt(const ThresholdArgs& a, uint32_t m, unsigned long long sx, unsigned long long sy,
                                                       bool& certain) {
    certain = true;
    const double md = (double)m, xd = (double)sx, yd = (double)sy;
    const double av = (md * yd - xd * xd) - xd * (md - 1.0);
    if (!(av > 0.0)) return false;  // c >= 0
    const double a2 = av * av, c2 = (a.nsig_b2 * (xd * xd)) * (2.0 * (md - 1.0));
    constexpr double kEps = 9.094947017729282e-13;  // 2^-40
    const bool yes = a2 > c2 + c2 * kEps, no = a2 < c2 - c2 * kEps;
    certain = yes || no;
    return yes;
}

__device__ __forceinline__ uint32_t dpp_shr_add(uint32_t from_neighbour, uint32_t addend) {
    // addend + (value of lane - 1); lane 0 adds 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)from_neighbour, 0x138, 0xf, 0xf, true) + addend;
}
__device__ __forceinline__ uint32_t dpp_shl_add(uint32_t from_neighbour, uint32_t addend) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)from_neighbour, 0x130, 0xf, 0xf, true) + addend;
}


// EXT = true: the first pass of the extended algorithm.  Same stream, ring, sums and queue; the group screen is a
// float32 upper bound of a = m y - x^2 - x (m - 1) against a lower bound of c (largest sum p^2 and smallest sum p of
// the group's eight windows), the drain decides "dispersion above background" exactly for every valid pixel of a
// queued group, and the plane written (its non-zero bytes; the host zeroes it before the launch) is a.dplane.
// DENSE: the byte mask is zero-filled too (somebody asked for it); a template parameter so that the hot path carries neither
// the stores nor their branch.
template <int KAHEAD, bool EXT = false, bool DENSE = false>
__global__ __launch_bounds__(64, 4) void k_stream_u16(const ThresholdArgs a) {
    __shared__ uint32_t s_q[kSQWords][kQCap];
    __shared__ uint16_t s_list[kQCap * 8];  // drain: (queue entry << 3 | pixel) of every candidate pixel

    const int lane = threadIdx.x;
    if (a.dbg_prio & 1) __builtin_amdgcn_s_setprio(3);   // (tuning "stream_prio": ahead of the band waves that share this SIMD in the issue arbitration)
    // XCD-aware unit map (ffs_device.h, stream_unit): a unit = one band of one strip, the strips of a band on one XCD
    int strip, band;
    const bool real = stream_unit(a, blockIdx.x, strip, band);
    if constexpr (!EXT)   // (the launch's last workgroup has started: every other one has been handed out -- see ThresholdArgs::handoff)
        if (a.handoff && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && lane == 0)
            __hip_atomic_store(a.handoff, a.handoff_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!real) return;   // (the grid is eight equal chunks: up to seven workgroups beyond the last unit)
    [[maybe_unused]] const uint32_t slot = log_slot(a, blockIdx.y, (uint32_t)band, (uint32_t)strip);   // where this wave's log lies
    const int f0 = blockIdx.y * a.group_frames;                   // first frame of this super row
    const int nf = min(a.group_frames, a.n_frames - f0);
    const int yb0 = band_first_row(band, a.band_rows, a.band_rows2, a.band_split);
    const int yb1 = min(band_first_row(band + 1, a.band_rows, a.band_rows2, a.band_split), a.H);

    // lane -> (frame, group) in the super row; group a.gpf of every frame is the empty separator
    const int gsep = a.gpf + 1;
    const int G = strip * kSOwned + lane - 1;
    const int fl = G >= 0 ? G / gsep : 0;
    const int g = G - fl * gsep;
    const bool active = G >= 0 && fl < nf && g < a.gpf;
    const bool owned = active && lane >= 1 && lane <= kSOwned;
    const uint32_t my_fg = ((uint32_t)fl << 16) | (uint32_t)g;   // travels with every group this lane queues

    const rsrc_t r_img = make_rsrc((const uint8_t*)a.image + (uint64_t)f0 * a.frame_stride,
                                   (uint32_t)((uint64_t)(nf - 1) * a.frame_stride + (uint64_t)a.H * a.pitch));
    const rsrc_t r_info = make_rsrc(a.ginfo, (uint32_t)(a.H + kInfoExtraRows) * a.gpitch);
    const rsrc_t r_sb = make_rsrc(a.strong_bytes + (uint64_t)f0 * a.bytes_frame_stride,
                                  (uint32_t)((uint64_t)nf * a.bytes_frame_stride));
    const rsrc_t r_cb = make_rsrc((EXT ? a.dplane : a.bits) + (uint64_t)f0 * a.plane_frame_stride, (uint32_t)((uint64_t)nf * a.plane_frame_stride));
    constexpr uint32_t kOob = 0x80000000u;  // offsets with bit 31 set are out of range for every resource (all < 2 GiB)
    const uint32_t off_px = active ? (uint32_t)((uint64_t)fl * a.frame_stride) + (uint32_t)g * 16u : kOob;
    const uint32_t off_info = active ? (uint32_t)g * 4u : kOob;
    // The byte mask is zero-filled in whole 128-byte lines, independent of who owns which pixel: the rows
    // of the nf frames hold nf * bpitch / 128 lines, wave `strip` clears lines 4 strip .. 4 strip + 3.
    uint32_t off_byte_st;
    {
        const uint32_t lpf = a.bpitch >> 7;                       // lines per frame row
        const uint32_t u = (uint32_t)strip * 4u + ((uint32_t)lane >> 4);
        const uint32_t fz = u / lpf, cz = u - fz * lpf;
        off_byte_st = fz < (uint32_t)nf ? (uint32_t)((uint64_t)fz * a.bytes_frame_stride) + cz * 128u + ((uint32_t)lane & 15u) * 8u : kOob;
    }

    const int total = (yb1 - yb0) + (FFS_DBG(a, 1024) ? 0 : 6);  // incoming rows yb0-3 .. yb1+2   (bit 1024: six rows fewer a wave -- a launch without warm-up rows; results are wrong)
    const float kS = a.kS;

    // Ring of NS = 7 + KAHEAD row slots, row i in slot i % NS: the seven rows of the window AND the rows in flight.  A row's
    // 16 bytes are loaded straight into its slot, masked there in place when the row comes in, and the slot of the row that
    // leaves the window (i - 7) is the one row i + KAHEAD is loaded into -- after the vertical update has read it.  No
    // register is ever copied (a seven-slot ring with the loads landing elsewhere cost four v_mov per row).
    constexpr int NS = 7 + KAHEAD;
    uint32_t ring[NS][4];  // masked pixels, two per register as loaded
    uint32_t rinfo[NS];    // ginfo dword that came with the row
    uint32_t col[8], colq[8];
    uint32_t M[4] = {0, 0, 0, 0};  // pair masks of the current mask byte
    uint32_t mprev = 0;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    [[maybe_unused]] f32x2 ext_c1 = {0.0f, 0.0f}, ext_c2 = {0.0f, 0.0f};   // EXT: screen constants of the current window counts, in pairs
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        rinfo[s] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[s][q] = 0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { col[j] = 0; colq[j] = 0; }
    // lanes that own output, as a wave mask (scalar): the flag ballot below is the compare's own mask ANDed with it
    const unsigned long long owned_mask = __builtin_amdgcn_ballot_w64(owned);

    // Loads of incoming row i.  Rows outside the image must count as zeros: their ginfo dword says "nothing valid" (the table's
    // rows H .. H + 2 hold no mask bits, rows above the image are killed below), so the pixels they bring are masked off in
    // mask_row like any masked pixel -- the pixel load itself only has to stay inside the buffer (row clamped: two scalar
    // instructions instead of a chain of compares and selects per load).
    auto fetch = [&](int slot, int i) {
        const int yin = yb0 - 3 + i;
        const uint32_t prow = (uint32_t)min(max(yin, 0), a.H - 1);
        const uint32_t irow = (uint32_t)min(max(yin, 0), a.H + kInfoExtraRows - 1);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_img, off_px, prow * a.pitch, FFS_IMG_LOAD_AUX);
        ring[slot][0] = v[0]; ring[slot][1] = v[1]; ring[slot][2] = v[2]; ring[slot][3] = v[3];
        rinfo[slot] = __builtin_amdgcn_raw_buffer_load_b32(r_info, off_info | (yin < 0 ? kOob : 0u), (FFS_DBG(a, 64) ? 8u : irow) * a.gpitch, 0);   // (bit 64: every row reads table row 8 -- what the table's traffic costs)
    };

    // the incoming row's four registers of two pixels each, masked in place
    auto mask_row = [&](int slot, int slot_before) {
        // the pair masks change only where the mask byte differs from the row above: module edges, dead pixels.  One compare
        // of the whole dwords tells "nothing changed" (the byte is part of it); only then is the byte itself looked at.
        if (__ballot(rinfo[slot] != rinfo[slot_before]) != 0ull) {  // wave-uniform; rare
            const uint32_t mb = rinfo[slot] & 0xFFu;
            if (__ballot(mb != mprev) != 0ull) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    M[q] = ((uint32_t)__builtin_amdgcn_sbfe((int)mb, 2 * q, 1) & 0xFFFFu)
                         | ((uint32_t)__builtin_amdgcn_sbfe((int)mb, 2 * q + 1, 1) & 0xFFFF0000u);
                mprev = mb;
            }
            if constexpr (EXT) {
                // the two constants of the first-pass screen follow the window counts that came with this row (see there):
                // taken here, where a change of the counts is noticed anyway, not once per row (a root and five more
                // instructions for values that change at module edges only)
                const uint32_t mmin = (rinfo[slot] >> 8) & 0xFFu, mmax = (rinfo[slot] >> 16) & 0xFFu;
                const float mm1 = (float)mmin - 1.0f;
                const float c1 = (float)mmax * 1.0000019073486328125f;
                const float c2 = (mm1 + a.kB * __builtin_amdgcn_sqrtf(2.0f * mm1)) * 0.99999952316284179688f;
                ext_c1 = f32x2{c1, c1};
                ext_c2 = f32x2{c2, c2};
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ring[slot][q] &= M[q];
            // keep the masked words opaque: otherwise the compiler merges `raw & M & 0xFFFF` into a three-operand
            // v_bitop3 per pair instead of taking the low half as an SDWA select of the word it already has
            asm("" : "+v"(ring[slot][q]));
        }
    };

    // vertical running sums: add the incoming row (slot s), retire the row that left the 7-row window (slot so).  The
    // half-word operands are meant to become SDWA selects (v_sub_u32_sdwa / v_add_u32_sdwa), the product a
    // v_mad_i32_i24: four instructions per pixel, nothing to unpack.
    // (Measured and dropped in round 3: no sum of squares in the stream at all, the ~0.6 % of pixels that pass the signal test
    // gathering their 49 pixels from L2 in the drain -- v_dot2_u32_u16 per pixel pair; 16 of 77 instructions per row less,
    // but every drain then waits for memory: 410 us per 32 Eiger frames against 311.)
    auto push = [&](int s, int so) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t wn = ring[s][q], wo = ring[so][q];
            {
                const int32_t t = (int32_t)((wn & 0xFFFFu) - (wo & 0xFFFFu));
                const int32_t u = (int32_t)((wn & 0xFFFFu) + (wo & 0xFFFFu));
                col[2 * q] += (uint32_t)t;
                colq[2 * q] += (uint32_t)__mul24(t, u);   // p_in^2 - p_out^2 (low 32 bits)
            }
            {
                const int32_t t = (int32_t)((wn >> 16) - (wo >> 16));
                const int32_t u = (int32_t)((wn >> 16) + (wo >> 16));
                col[2 * q + 1] += (uint32_t)t;
                colq[2 * q + 1] += (uint32_t)__mul24(t, u);
            }
        }
    };

    int qn = 0;  // queued lane groups (wave-uniform)
    [[maybe_unused]] uint32_t nlog = 0;   // entries of this wave's log already in memory (wave-uniform)
    [[maybe_unused]] uint32_t nbuf = 0;   // ... and still in lbuf
    [[maybe_unused]] uint32_t lbuf[6] = {0, 0, 0, 0, 0, 0};   // lane e: entry nlog + e (two words) and its group's eight pixels (four)
    [[maybe_unused]] auto flush_log = [&]() {
        const uint32_t at = nlog + (uint32_t)lane;
        if ((uint32_t)lane < nbuf && at < (uint32_t)kWlogCap && !FFS_DBG(a, 512)) {   // (bit 512: the log stays unwritten and empty -- what its stores cost)
            a.wlog[(uint64_t)slot * kWlogCap + at] = make_uint2(lbuf[0], lbuf[1]);
            a.wpix[(uint64_t)slot * kWlogCap + at] = make_uint4(lbuf[2], lbuf[3], lbuf[4], lbuf[5]);
        }
        nlog += nbuf;
        nbuf = 0;
    };
    // 64 queued groups at a time.  Phase 1, one group per lane: the conservative float32 signal test on its
    // eight pixels (a proven superset, see signal_test8).  Phase 2: the pixels still standing (a few dozen)
    // are dealt one per lane -- each lane's candidates numbered by ballot + mbcnt, bit plane by bit plane --
    // and take the oracle's predicate on their exact window sums side by side instead of one after the
    // other inside their group's lane (the drain is a latency chain, not a throughput problem).
    // Phase 3: the group's lane collects its result byte.
    auto drain = [&]() {
        const bool have = lane < qn && !FFS_DBG(a, 2) && !FFS_DBG(a, 32);   // (bit 32 leaves the queue unwritten: nothing to read back)
        uint32_t todo = 0, row = 0, fe = 0, ge = 0;
        if (have) {
            // everything phase 1 reads of the entry is asked for HERE, in one LDS round trip: left where it is used, the window sums and
            // the pixels were fetched behind the branch on the group's counts -- a second round trip in every drain (section 3.2e)
            uint32_t xs[8], pws[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[j] = s_q[j][lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) pws[q] = s_q[8 + q][lane];
            row = s_q[30][lane] >> 6;
            const uint32_t fg = s_q[31][lane];   // (the pushing lane's own frame and group: no division here)
            fe = fg >> 16;
            ge = fg & 0xFFFFu;
            // window counts of the eight pixels: one value for the whole group almost everywhere (then ginfo has it);
            // only groups next to masked pixels fetch their eight counts -- a dependent global round trip at the head
            // of the drain that most waves now never pay
            const uint32_t ginf = s_q[15][lane];
            const uint32_t gmin = (ginf >> 8) & 0xFFu, gmax = (ginf >> 16) & 0xFFu;
            uint2 mm = make_uint2(gmin * 0x01010101u, gmin * 0x01010101u);
            // (row and group clamped into the table: this is the one access of the kernel that is a plain global load at an address
            // that comes out of the LDS queue -- a tag that is not what a push wrote, as under round 3's experiment bit 32 which left
            // the queue unwritten, must not become a wild address: DESIGN.md section 10.  A buffer resource for it cost four SGPRs
            // over the whole kernel, 17 spills instead of 13 and 1-3 % of the step.)
            if (gmin != gmax) {
                uint2 t = *reinterpret_cast<const uint2*>(a.mmap + (uint64_t)min(row, (uint32_t)a.H - 1u) * a.pitch_px + min(ge * 8u, (uint32_t)a.pitch_px - 8u));
                // The load is waited for HERE, inside the rare branch.  Left to the compiler the wait stood behind the branch's join --
                // `s_waitcnt vmcnt(0)` in EVERY drain, load or no load -- and vmcnt counts the row loop's prefetched pixel rows too: each
                // drain waited until the three rows in flight had all arrived.
                asm volatile("" : "+v"(t.x), "+v"(t.y));
                mm = t;
            }
            s_q[12][lane] = mm.x;
            s_q[13][lane] = mm.y;
            s_q[14][lane] = 0u;
            uint32_t sure = 0;   // EXT: pixels the float32 form already proves "above background"
            if constexpr (EXT) {
                // every VALID pixel of the group with enough pixels in its window (and a centre not above max_valid,
                // the device kernels' rule) is decided here where float32 can: with t0 = m y, the float32 value af of
                // a = t0 - x (x + m - 1) is within 2^-22 t0 of the true one (y and the two products rounded once each, x and
                // x + m - 1 exact below 2^24, the difference once), cf = x kB sqrt(2 (m - 1)) within 2^-19 of the true c
                // (kB = nsig_b (1 - 2^-20), one ulp each for the root and the two products).  2^-20 t0 and 2^-18 cf are granted:
                //   af + 2^-20 t0 <  cf               -> certainly not above background
                //   af - 2^-20 t0 >  cf (1 + 2^-18)   -> certainly above (float64 roundings of the oracle: 2^-50 t0, far inside)
                // and only the band between the two takes the float64 test of phase 2 (a few pixels per million).
                const uint32_t valid = ginf >> 24;
                // one count for the whole group almost everywhere: its root is taken once
                const float kq_group = a.kB * __builtin_amdgcn_sqrtf(2.0f * ((float)gmin - 1.0f));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t x = xs[j];
                    const uint32_t pw = pws[j >> 1];
                    const uint32_t pv = (j & 1) ? pw >> 16 : pw & 0xFFFFu;
                    const uint32_t m = ((j < 4 ? mm.x : mm.y) >> (8 * (j & 3))) & 0xFFu;
                    const float mf = (float)m, xf = (float)x, yf = (float)s_q[16 + j][lane];   // (the window's sum of p^2, as pushed)
                    const float kq = gmin == gmax ? kq_group : a.kB * __builtin_amdgcn_sqrtf(2.0f * (mf - 1.0f));
                    const float t0 = mf * yf;
                    const float af = t0 - xf * (xf + (mf - 1.0f));
                    const float cf = xf * kq;
                    const float slack = t0 * 9.5367431640625e-07f;
                    const bool bright = x >= 65536u;   // (the 32-bit sums of p^2 may have wrapped: phase 2 hands it to k_bright_fix)
                    const bool maybe = (af + slack >= cf) || bright;
                    const bool yes = !bright && (af - slack > cf * 1.000003814697265625f);
                    const bool ok = ((valid >> j) & 1u) && (int)m >= a.min_count && !(a.max_valid >= 0 && (long long)pv > a.max_valid);
                    todo |= (ok && maybe && !yes) ? (1u << j) : 0u;
                    sure |= (ok && yes) ? (1u << j) : 0u;
                }
                s_q[14][lane] = sure;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t x = xs[j];
                    const uint32_t pw = pws[j >> 1];
                    const uint32_t pv = (j & 1) ? pw >> 16 : pw & 0xFFFFu;
                    const uint32_t m = ((j < 4 ? mm.x : mm.y) >> (8 * (j & 3))) & 0xFFu;
                    //   oracle: b = m p - x > nsig_s sqrt(x m);  here: b |b| > nsig_s^2 (1 - 2^-16) x m in float32
                    const int32_t b = (int32_t)__umul24(m, pv) - (int32_t)x;
                    const float bf = (float)b, tf = (float)__umul24(x, m);
                    todo |= (bf * __builtin_fabsf(bf) > kS * tf) ? (1u << j) : 0u;
                }
            }
            if (FFS_DBG(a, 4)) todo = 0;
        }
        // the candidates listed densely (any order will do): a scan of the lanes' counts, then every lane writes its own -- as many
        // rounds as the busiest lane has candidates (one or two) instead of eight ballots
        const uint32_t todo_n = (uint32_t)__popc(todo);
        const uint32_t todo_incl = wave_inclusive_scan(todo_n);
        const int T = __builtin_amdgcn_readlane((int)todo_incl, 63);  // candidates of the whole wave (wave-uniform)
        {
            uint32_t pos = todo_incl - todo_n, w = todo;
            while (w) {
                const uint32_t j = (uint32_t)__ffs((int)w) - 1u;
                w &= w - 1u;
                s_list[pos++] = (uint16_t)(((uint32_t)lane << 3) | j);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < T; base += 64) {  // wave-uniform; one round unless the groups are full of bright pixels
            const int idx = base + lane;
            if (idx < T) {
                const uint32_t ent = s_list[idx], e = ent >> 3, j = ent & 7u;
                // (the candidate's ten words in ONE round trip: read where they were used, x came first, the branch on it next, and the
                // column sums behind that -- three round trips)
                const uint32_t x = s_q[j][e], pv = (s_q[8 + (j >> 1)][e] >> (16 * (j & 1))) & 0xFFFFu;
                const uint32_t m = (s_q[12 + (j >> 2)][e] >> (8 * (j & 3))) & 0xFFu;
                // sum p^2 mod 2^32 is the true sum while x < 65536 (y <= 65535 x < 2^32); window j = cq[j .. j+6]
                uint32_t y = 0;
                if constexpr (EXT) {
                    y = s_q[16 + j][e];
                } else {
#pragma unroll
                    for (uint32_t t = 0; t < 7; ++t) y += s_q[16 + j + t][e];
                }
                asm volatile("" : "+v"(y));   // (keeps the reads ahead of the branch)
                if (x < 65536u) {
                    bool certain;
                    bool strong;
                    if constexpr (EXT) {
                        strong = dispersion_only_nosqrt(a, m, x, y, certain);
                        if (!certain) strong = dispersion_only(a, m, x, y);
                    } else {
                        if (a.int_pred) strong = int_predicate(a, m, x, y, pv, certain);   // (wave-uniform)
                        else strong = exact_predicate_nosqrt(a, m, x, y, pv, certain);
                        if (!certain) strong = exact_predicate(a, m, x, y, pv);
                    }
                    if (strong) atomicOr(&s_q[14][e], 1u << j);
                } else if (!EXT && a.bright_to_plane) {
                    // sum p^2 may not fit 32 bits: marked as a candidate, k_exact gathers the window and decides
                    atomicOr(&s_q[14][e], 1u << j);
                } else if (!EXT && a.wlog) {
                    // ... marked "undecided" in the group's log entry: the sparse launch gathers the window and decides
                    atomicOr(&s_q[14][e], 0x100u << j);
                } else {
                    // sum p^2 may not fit 32 bits: k_bright_fix gathers the window and decides (rare)
                    const uint32_t tg = s_q[30][e], fgc = s_q[31][e];
                    const uint32_t fc = fgc >> 16, gc = fgc & 0xFFFFu;
                    const uint32_t at = atomicAdd(a.bright_n, 1u);
                    if (at < a.bright_cap)
                        a.bright_list[at] = make_uint2(((uint32_t)(f0 + (int)fc) << 16) | (gc * 8u + j), tg >> 6);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (!EXT) if (a.wlog) {   // (wave-uniform) one 24-byte entry per group instead of a plane byte and two atomics
            // The entries wait in REGISTERS (entry e of the current run of 64 in lane e: lbuf) and leave the wave 64 at a time
            // and when it ends, as whole lines.  Stored drain by drain (5 entries, two store instructions) they cost the kernel
            // 12 us of 292 (tools/log_store_probe.py: the same kernel with the log left unwritten) -- 120 k partial-line writes
            // whose completion the row loads behind them are counted after (vmcnt is one in-order counter on gfx9).
            const uint32_t cbm = have ? s_q[14][lane] : 0u;   // strong | undecided << 8
            const uint32_t pxw[4] = {s_q[8][lane], s_q[9][lane], s_q[10][lane], s_q[11][lane]};   // the group's pixels (asked for with cbm: one round trip)
            const unsigned long long wm = __builtin_amdgcn_ballot_w64(cbm != 0u);
            if (wm) {
                const uint32_t cnt = (uint32_t)__popcll(wm);
                if (nbuf + cnt > 64u) flush_log();
                // every lane with an entry sends its six words to the lane that will hold them (ds_permute_b32: the crossbar, no LDS
                // memory and no second round trip); the lanes without one send to a lane outside the run's new part, which ignores it
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u));
                const uint32_t dump = nbuf + cnt < 64u ? nbuf + cnt : 0u;
                const int to = (int)((cbm != 0u ? nbuf + rank : dump) << 2);
                const uint32_t e0 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)((row << 16) | ge));
                const uint32_t e1 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)((fe << 16) | cbm));
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[0]), p1 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[1]);
                const uint32_t p2 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[2]), p3 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[3]);
                if ((uint32_t)lane - nbuf < cnt) {
                    lbuf[0] = e0; lbuf[1] = e1;
                    lbuf[2] = p0; lbuf[3] = p1; lbuf[4] = p2; lbuf[5] = p3;   // the group's pixels
                }
                nbuf += cnt;   // (queue entries are in (row, lane) order, so the log is sorted by (row, frame, group))
            }
            qn = 0;
            return;
        }
        if (have) {
            const uint32_t cb = s_q[14][lane];
            if (cb) {  // the plane is all zero when the kernel starts (the compaction clears what it consumed)
                if (!FFS_DBG(a, 256)) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)cb, r_cb, (uint32_t)((uint64_t)fe * a.plane_frame_stride) + ge, row * a.mpitch, 0);
                if constexpr (!EXT) if (!FFS_DBG(a, 128)) {  // (the extended algorithm's final pass counts its own strong pixels)
                    atomicAdd(a.tile_counts + (uint64_t)(f0 + (int)fe) * a.n_tiles + (row / (uint32_t)kTileRows), (uint32_t)__popc(cb));
                    const uint32_t ob = row * a.occ_spr + (ge >> 4);   // the 16-byte segment of the plane row this byte lies in
                    atomicOr(a.occ + (uint64_t)(f0 + (int)fe) * a.occ_frame_words + (ob >> 5), 1u << (ob & 31u));
                }
            }
        }
        qn = 0;
    };

#pragma unroll
    for (int s = 0; s < KAHEAD; ++s) fetch(s, s);

    // warm-up: rows 0..5 of the band's input only fill the window (their "outgoing" slots hold zeros)
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        mask_row(s, (s + NS - 1) % NS);   // (row 0: against the zeroed slot before it)
        push(s, (s + KAHEAD) % NS);
        fetch((s + KAHEAD) % NS, s + KAHEAD);
    }

    for (int base = 6;; base += NS) {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int s = (6 + t) % NS;        // slot of incoming row i (i % NS == s)
            const int so = (s + KAHEAD) % NS;  // slot of the row that leaves the window, i - 7; row i + KAHEAD is loaded into it
            const int sc = (s + NS - 3) % NS;  // slot of the centre row i - 3
            const int i = base + t;
            if (i >= total) goto rows_done;
            {
                const uint32_t info = rinfo[s];   // carries the counts of the centre row
                mask_row(s, (s + NS - 1) % NS);
                push(s, so);
                fetch(so, i + KAHEAD);

                // horizontal 7-tap over the column sums c[-3..10] = L5 L6 L7 c0..c7 R0 R1 R2, from pair sums;
                // every neighbour value rides on an add
                const uint32_t s01 = col[0] + col[1], s23 = col[2] + col[3], s45 = col[4] + col[5], s67 = col[6] + col[7];
                const uint32_t ua = s01 + s23, va = s45 + s67;
                const uint32_t T0 = dpp_shr_add(s67, ua);      // L6 L7 c0..c3
                const uint32_t T1 = ua + s45;                   // c0..c5
                const uint32_t T2 = va + s23;                   // c2..c7
                const uint32_t T3 = dpp_shl_add(s01, va);      // c4..c7 R0 R1
                uint32_t Wn[8];
                Wn[0] = dpp_shr_add(col[5], T0);
                Wn[1] = T0 + col[4];
                Wn[2] = dpp_shr_add(col[7], T1);
                Wn[3] = T1 + col[6];
                Wn[4] = T2 + col[1];
                Wn[5] = dpp_shl_add(col[0], T2);
                Wn[6] = T3 + col[3];
                Wn[7] = dpp_shl_add(col[2], T3);

                const int yout = __builtin_amdgcn_readfirstlane(yb0 + (i - 6));
                const uint32_t so_bytes = (uint32_t)yout * a.bpitch;

                // group screen: for every valid pixel j of the group  b_j = m_j p_j - x_j <= mmax pmax - xmin  and
                // nsig_s sqrt(x_j m_j) >= nsig_s sqrt(xmin mmin): a group that fails  B |B| > kS xmin mmin  has no
                // candidate (kS = nsig_s^2 (1 - 2^-16): float32 rounding cannot turn a true pass into a fail)
                // (three-operand chains: four v_min3 / v_max3 class instructions each instead of five)
                const uint32_t xmin = min(min(min(min(Wn[0], Wn[1]), Wn[2]), min(min(Wn[3], Wn[4]), Wn[5])), min(Wn[6], Wn[7]));
                const uint32_t mmin = (info >> 8) & 0xFFu, mmax = (info >> 16) & 0xFFu;
                bool pass;
                [[maybe_unused]] uint32_t ys[8];   // EXT: the eight windows' sums of p^2 (the queue takes them as they are)
                if constexpr (EXT) {
                    // For every valid pixel j of the group:  a_j = m_j y_j - x_j^2 - x_j (m_j - 1) <= mmax y_j - x_j (x_j + mmin - 1)  and
                    // c_j = nsig_b x_j sqrt(2 (m_j - 1)) >= nsig_b x_j sqrt(2 (mmin - 1)).  The eight windows of sum p^2 slide over
                    // the 14 column sums as the sums of p do; float32 with the allowances of group_tests8 (|fl(a) - a| < 2^-21 m y;
                    // 2^-20 m y granted, kB = nsig_b (1 - 2^-20)).  The 32-bit sums of p^2 are the true ones while every window
                    // sum is below 65536; brighter groups pass.
                    const uint32_t q01 = colq[0] + colq[1], q23 = colq[2] + colq[3], q45 = colq[4] + colq[5], q67 = colq[6] + colq[7];
                    const uint32_t qa = q01 + q23, qb = q45 + q67;
                    const uint32_t U0 = dpp_shr_add(q67, qa), U1 = qa + q45, U2 = qb + q23, U3 = dpp_shl_add(q01, qb);
                    const uint32_t y0 = dpp_shr_add(colq[5], U0), y1 = U0 + colq[4], y2 = dpp_shr_add(colq[7], U1), y3 = U1 + colq[6];
                    const uint32_t y4 = U2 + colq[1], y5 = dpp_shl_add(colq[0], U2), y6 = U3 + colq[3], y7 = dpp_shl_add(colq[2], U3);
                    // per pixel (a bound per GROUP pairs the largest sum p^2 with the smallest sum p and lets most of the
                    // background through): d_j = [mmax y_j - x_j (x_j + mmin - 1)] (1 + 2^-20) - kB x_j sqrt(2 (mmin - 1)),
                    // a pixel can only pass the first pass where d_j >= 0; the lane keeps the largest d_j
                    ys[0] = y0; ys[1] = y1; ys[2] = y2; ys[3] = y3; ys[4] = y4; ys[5] = y5; ys[6] = y6; ys[7] = y7;
                    // d_j = c1 y_j - x_j (x_j + c2) with c1 = mmax (1 + 2^-19) and c2 = (mmin - 1 + kB sqrt(2 (mmin - 1))) (1 - 2^-21): an upper
                    // bound of a_j - c_j in float32 whatever the roundings do (y_j, x_j + c2, the product and the fused difference:
                    // each within 2^-24 of a term no larger than c1 y_j or x_j (x_j + c2); the two factors grant 2^-19 and 2^-21 of
                    // them).  Two pixels per instruction: v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32 -- 12 + 16 conversions + 4
                    // v_max3 per row (the compiler's own packing of the one-pixel form spent 19 v_mov on pairing registers).
                    float dmax = -1.0f;
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        const f32x2 xf = {(float)Wn[2 * jp], (float)Wn[2 * jp + 1]};
                        const f32x2 yf = {(float)ys[2 * jp], (float)ys[2 * jp + 1]};
                        const f32x2 u = xf * (xf + ext_c2);
                        const f32x2 d = __builtin_elementwise_fma(ext_c1, yf, -u);
                        dmax = __builtin_fmaxf(dmax, __builtin_fmaxf(d.x, d.y));
                    }
                    const uint32_t xmax = max(max(max(max(Wn[0], Wn[1]), Wn[2]), max(max(Wn[3], Wn[4]), Wn[5])), max(Wn[6], Wn[7]));
                    pass = mmax != 0u && (dmax >= 0.0f || xmax >= 65536u);
                } else {
                    uint32_t pmax;
                    {
                        typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
                        const u16x2 a01 = __builtin_bit_cast(u16x2, ring[sc][0]), a23 = __builtin_bit_cast(u16x2, ring[sc][1]);
                        const u16x2 a45 = __builtin_bit_cast(u16x2, ring[sc][2]), a67 = __builtin_bit_cast(u16x2, ring[sc][3]);
                        const u16x2 mx = __builtin_elementwise_max(__builtin_elementwise_max(a01, a23), __builtin_elementwise_max(a45, a67));  // v_pk_max_u16
                        const uint32_t mw = __builtin_bit_cast(uint32_t, mx);
                        pmax = max(mw & 0xFFFFu, mw >> 16);
                    }
                    const int32_t B = (int32_t)__umul24(mmax, pmax) - (int32_t)xmin;
                    const float bf = (float)B, tf = (float)__umul24(xmin, mmin);
                    float lhs = bf * __builtin_fabsf(bf);
                    asm("" : "+v"(lhs));   // (keeps the two products apart: packed, v_pk_mul_f32 costs a v_mov and a double-rate slot)
                    pass = lhs > kS * tf;
                }
                if constexpr (DENSE) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, r_sb, off_byte_st, so_bytes, 2 /* nt: written once, read much later */);
                // (the compare's mask ANDed with a scalar: no VALU instruction turns the flag back into a mask)
                const unsigned long long fm = FFS_DBG(a, 1) ? 0ull : (__builtin_amdgcn_ballot_w64(pass) & owned_mask);
                const bool flag = pass && owned;
                if (fm) {  // wave-uniform
                    const int nfl = __popcll(fm);
                    if (qn + nfl > kQCap) drain();
                    if constexpr (EXT) {
                        // the first pass's screen has the eight windows' sums of p^2 in hand: they go into the queue as they are
                        // (words 16-23) -- no neighbour exchange, six words less, and the drain does not slide windows again
                        if (flag && !FFS_DBG(a, 32)) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                            const int e = qn + (int)rank;   // < kQCap: after a drain qn = 0 and nfl <= 62
#pragma unroll
                            for (int w = 0; w < 8; ++w) {
                                s_q[w][e] = Wn[w];
                                s_q[16 + w][e] = ys[w];
                            }
#pragma unroll
                            for (int w = 0; w < 4; ++w) s_q[8 + w][e] = ring[sc][w];
                            s_q[15][e] = info;
                            s_q[30][e] = ((uint32_t)yout << 6) | (uint32_t)lane; s_q[31][e] = my_fg;
                        }
                    } else {
                    // the group's 14 column sums of p^2 (own 8 + 3 from each neighbour lane)
                    const uint32_t QL5 = from_left(colq[5]), QL6 = from_left(colq[6]), QL7 = from_left(colq[7]);
                    const uint32_t QR0 = from_right(colq[0]), QR1 = from_right(colq[1]), QR2 = from_right(colq[2]);
                    if (flag && !FFS_DBG(a, 32)) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        const int e = qn + (int)rank;   // < kQCap: after a drain qn = 0 and nfl <= 62
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            s_q[w][e] = Wn[w];
                            s_q[19 + w][e] = colq[w];
                        }
#pragma unroll
                        for (int w = 0; w < 4; ++w) s_q[8 + w][e] = ring[sc][w];
                        s_q[15][e] = info;
                        s_q[16][e] = QL5; s_q[17][e] = QL6; s_q[18][e] = QL7;
                        s_q[27][e] = QR0; s_q[28][e] = QR1; s_q[29][e] = QR2;
                        s_q[30][e] = ((uint32_t)yout << 6) | (uint32_t)lane; s_q[31][e] = my_fg;
                    }
                    }
                    qn += nfl;
                }
            }
        }
    }
rows_done:
    if (qn > 0) drain();
    if constexpr (!EXT) if (a.wlog) {
        flush_log();
        if (lane == 0) a.wlog_n[slot] = FFS_DBG(a, 512) ? 0u : nlog;
    }
}
// Pixels whose window holds sum p >= 65536 (k_stream_u16 cannot vouch for its 32-bit sum of p^2): exact
// 64-bit sums gathered from memory, then the same predicate.  A handful per frame at most.
template <typename PixelT, bool EXT = false>
__global__ __launch_bounds__(256) void k_bright_fix(const ThresholdArgs a) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && *a.bright_n > a.bright_cap) atomicOr(a.overflow, 8u);  // the host re-runs the batch
    const uint32_t n = min(*a.bright_n, a.bright_cap);
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const uint2 r = a.bright_list[e];
        const uint32_t frame = r.x >> 16, x = r.x & 0xFFFFu, y = r.y;
        const uint8_t* img = (const uint8_t*)a.image + (uint64_t)frame * a.frame_stride;
        if (exact_strong<PixelT, EXT>(a, img, (int)x, (int)y)) {  // EXT: the dispersion half alone, into the first-pass plane
            uint8_t* plane = (EXT ? a.dplane : a.bits) + (uint64_t)frame * a.plane_frame_stride + (uint64_t)y * a.mpitch;
            atomicOr(reinterpret_cast<uint32_t*>(plane) + (x >> 5), 1u << (x & 31u));  // rows start on 4-byte boundaries
            if (!EXT) {
                atomicAdd(a.tile_counts + (uint64_t)frame * a.n_tiles + y / (uint32_t)kTileRows, 1u);
                const uint32_t ob = y * a.occ_spr + (x >> 7);
                atomicOr(a.occ + (uint64_t)frame * a.occ_frame_words + (ob >> 5), 1u << (ob & 31u));
            }
        }
    }
}


// ================================================================================================================
// 32-bit pixels (the reference's PIXEL_DATA_32BIT build, h5read.h:16-20): the same one-kernel formulation with four
// pixels (16 bytes) per lane.  Differences from k_stream_u16: the ring holds the four masked pixels of a row as they
// are (nothing packed), p_in^2 - p_out^2 is a 32-bit multiply (low word: the sums are exact whenever they are used,
// i.e. while the window sum is below 65536), a lane pair shares one byte of the strong plane (results are ORed in),
// and the oracle's `src < 2^24` rule: a window that holds a valid pixel >= 2^24 has other sums AND another count
// than the mask alone gives, so every valid pixel of a group whose 7-row x 10-column neighbourhood holds one goes
// to the gather path (k_bright_fix<uint32_t>) -- never taken on photon-counting data, there for exactness.
constexpr int kS4Words = 32;  // queue entry: 0-3 window sums, 4-7 centre pixels, 8 ginfo, 9 window counts, 10 result bits,
                              // 11 tag (bit 31: neighbourhood holds a pixel >= 2^24), 12-21 / 22-31 low / high words of the
                              // column sums of p^2 (L1 L2 L3 c0..c3 R0 R1 R2)

template <int KAHEAD, bool DENSE = false>
__global__ __launch_bounds__(64, 4) void k_stream_u32(const ThresholdArgs a) {
    __shared__ uint32_t s_q[kS4Words][kQCap];
    __shared__ uint16_t s_list[kQCap * 4];

    const int lane = threadIdx.x;
    int strip, band;   // (the unit map: see k_stream_u16)
    const bool real = stream_unit(a, blockIdx.x, strip, band);
    if (a.handoff && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && lane == 0)
        __hip_atomic_store(a.handoff, a.handoff_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!real) return;
    const uint32_t slot = log_slot(a, blockIdx.y, (uint32_t)band, (uint32_t)strip);
    const int f0 = blockIdx.y * a.group_frames;
    const int nf = min(a.group_frames, a.n_frames - f0);
    const int yb0 = band_first_row(band, a.band_rows, a.band_rows2, a.band_split);
    const int yb1 = min(band_first_row(band + 1, a.band_rows, a.band_rows2, a.band_split), a.H);

    const int gsep = a.gpf + 1;   // a.gpf = groups of four pixels per frame row
    const int G = strip * kSOwned + lane - 1;
    const int fl = G >= 0 ? G / gsep : 0;
    const int g = G - fl * gsep;
    const bool active = G >= 0 && fl < nf && g < a.gpf;
    const bool owned = active && lane >= 1 && lane <= kSOwned;

    const rsrc_t r_img = make_rsrc((const uint8_t*)a.image + (uint64_t)f0 * a.frame_stride,
                                   (uint32_t)((uint64_t)(nf - 1) * a.frame_stride + (uint64_t)a.H * a.pitch));
    const rsrc_t r_info = make_rsrc(a.ginfo, (uint32_t)(a.H + kInfoExtraRows) * a.gpitch);
    const rsrc_t r_sb = make_rsrc(a.strong_bytes + (uint64_t)f0 * a.bytes_frame_stride,
                                  (uint32_t)((uint64_t)nf * a.bytes_frame_stride));
    constexpr uint32_t kOob = 0x80000000u;
    const uint32_t off_px = active ? (uint32_t)((uint64_t)fl * a.frame_stride) + (uint32_t)g * 16u : kOob;
    const uint32_t off_info = active ? (uint32_t)g * 4u : kOob;
    uint32_t off_byte_st;   // zero-fill: wave `strip` clears the 128-byte lines 2 strip, 2 strip + 1 of the super row
    {
        const uint32_t lpf = a.bpitch >> 7;
        const uint32_t u = (uint32_t)strip * 2u + ((uint32_t)lane >> 4);
        const uint32_t fz = u / lpf, cz = u - fz * lpf;
        off_byte_st = (lane < 32 && fz < (uint32_t)nf) ? (uint32_t)((uint64_t)fz * a.bytes_frame_stride) + cz * 128u + ((uint32_t)lane & 15u) * 8u : kOob;
    }

    const int total = (yb1 - yb0) + 6;
    const float kS = a.kS;

    // ring of NS = 7 + KAHEAD row slots, loads landing in their slot and masked in place (see k_stream_u16)
    constexpr int NS = 7 + KAHEAD;
    uint32_t ring[NS][4];
    uint32_t rinfo[NS];
    uint32_t col[4];
    long long colq[4];   // sum p^2 over the 7 rows, exact (49 * 2^48 < 2^63)
    uint32_t M[4] = {0, 0, 0, 0};  // 0 / ~0 per pixel of the current mask nibble
    uint32_t mprev = 0;
    uint32_t bighist = 0;          // bit s: the row in ring slot s (a row of the window) holds a valid pixel >= 2^24 in this lane
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        rinfo[s] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[s][q] = 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { col[j] = 0; colq[j] = 0; }
    const unsigned long long owned_mask = __builtin_amdgcn_ballot_w64(owned);

    auto fetch = [&](int slot, int i) {   // (rows outside the image: masked off through their ginfo dword, see k_stream_u16)
        const int yin = yb0 - 3 + i;
        const uint32_t prow = (uint32_t)min(max(yin, 0), a.H - 1);
        const uint32_t irow = (uint32_t)min(max(yin, 0), a.H + kInfoExtraRows - 1);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_img, off_px, prow * a.pitch, FFS_IMG_LOAD_AUX);
        ring[slot][0] = v[0]; ring[slot][1] = v[1]; ring[slot][2] = v[2]; ring[slot][3] = v[3];
        rinfo[slot] = __builtin_amdgcn_raw_buffer_load_b32(r_info, off_info | (yin < 0 ? kOob : 0u), irow * a.gpitch, 0);
    };

    auto mask_row = [&](int slot, int slot_before) {
        if (__ballot(rinfo[slot] != rinfo[slot_before]) != 0ull) {  // wave-uniform; rare
            const uint32_t mb = rinfo[slot] & 0xFu;
            if (__ballot(mb != mprev) != 0ull) {
#pragma unroll
                for (int q = 0; q < 4; ++q) M[q] = (uint32_t)__builtin_amdgcn_sbfe((int)mb, q, 1);
                mprev = mb;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[slot][q] &= M[q];
    };

    auto push = [&](int s, int so) {
        const uint32_t big = ((ring[s][0] | ring[s][1]) | (ring[s][2] | ring[s][3])) >> 24 ? 1u : 0u;
        bighist = (bighist & ~((1u << s) | (1u << so))) | (big << s);   // (slot so leaves the window)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int32_t t = (int32_t)(ring[s][q] - ring[so][q]), u = (int32_t)(ring[s][q] + ring[so][q]);  // |t|, u < 2^25 (a pixel >= 2^24 sends its windows to the gather path)
            col[q] += (uint32_t)t;
            colq[q] += (long long)t * (long long)u;   // p_in^2 - p_out^2: one v_mad_i64_i32
        }
    };

    int qn = 0;
    uint32_t nlog = 0;   // entries of this wave's log already in memory (wave-uniform; ThresholdArgs::wlog)
    uint32_t nbuf = 0;   // ... and still in lbuf: entry nlog + e in lane e, written out 64 at a time and when the wave ends (see k_stream_u16)
    uint32_t lbuf[6] = {0, 0, 0, 0, 0, 0};
    auto flush_log = [&]() {
        const uint32_t at = nlog + (uint32_t)lane;
        if ((uint32_t)lane < nbuf && at < (uint32_t)kWlogCap) {
            a.wlog[(uint64_t)slot * kWlogCap + at] = make_uint2(lbuf[0], lbuf[1]);
            a.wpix[(uint64_t)slot * kWlogCap + at] = make_uint4(lbuf[2], lbuf[3], lbuf[4], lbuf[5]);
        }
        nlog += nbuf;
        nbuf = 0;
    };
    auto drain = [&]() {
        const bool have = lane < qn && !FFS_DBG(a, 2) && !FFS_DBG(a, 32);   // (bit 32 leaves the queue unwritten, as in k_stream_u16: nothing to read back)
        uint32_t todo = 0, row = 0, fe = 0, ge = 0, info = 0;
        bool big = false;
        if (have) {
            const uint32_t tag = s_q[11][lane], ln = tag & 63u;
            big = (tag >> 31) != 0u;
            row = (tag & 0x7FFFFFFFu) >> 6;
            const int Ge = strip * kSOwned + (int)ln - 1;
            fe = (uint32_t)Ge / (uint32_t)gsep;
            ge = (uint32_t)Ge - fe * (uint32_t)gsep;
            info = s_q[8][lane];
            uint32_t xs[4], pvs[4];   // (asked for here, with the tag and the counts' bounds: one LDS round trip -- see k_stream_u16)
#pragma unroll
            for (int j = 0; j < 4; ++j) { xs[j] = s_q[j][lane]; pvs[j] = s_q[4 + j][lane]; }
            // window counts of the four pixels: one value for the whole group almost everywhere (then ginfo has it, as in k_stream_u16);
            // only groups next to masked pixels fetch their counts -- until round 5 EVERY drain of this kernel began with this dependent
            // global round trip (tag -> division -> address -> load)
            const uint32_t gmin = (info >> 8) & 0xFFu, gmax = (info >> 16) & 0xFFu;
            uint32_t mm = gmin * 0x01010101u;
            if (gmin != gmax) {
                uint32_t t = *reinterpret_cast<const uint32_t*>(a.mmap + (uint64_t)min(row, (uint32_t)a.H - 1u) * a.pitch_px
                                                                + min(ge * 4u, (uint32_t)a.pitch_px - 4u));   // (clamped: see k_stream_u16)
                asm volatile("" : "+v"(t));   // (waited for inside the branch, not behind its join: see k_stream_u16)
                mm = t;
            }
            s_q[9][lane] = mm;
            s_q[10][lane] = 0u;
            if (!big) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t x = xs[j], pv = pvs[j];
                    const uint32_t m = (mm >> (8 * j)) & 0xFFu;
                    // conservative float32 signal test (b exact in 32 bits; the conversions stay far inside the 2^-16 margin of kS)
                    const int32_t b = (int32_t)(m * pv) - (int32_t)x;
                    const float bf = (float)b, tf = (float)x * (float)m;
                    todo |= (bf * __builtin_fabsf(bf) > kS * tf) ? (1u << j) : 0u;
                }
            }
        }
        // groups next to a pixel >= 2^24: every valid pixel to the gather path
        if (__ballot(big) != 0ull) {
            if (big && a.bright_to_plane) {
                s_q[10][lane] = (info >> 24) & 0xFu;   // every valid pixel of the group is a candidate for k_exact
            } else if (big && a.wlog) {
                s_q[10][lane] = ((info >> 24) & 0xFu) << 8;   // ... "undecided" in the group's log entry: the sparse launch gathers and decides
            } else if (big) {
                for (uint32_t j = 0; j < 4; ++j) {
                    if (!((info >> (24 + j)) & 1u)) continue;   // (byte 3 = mask bits of the centre row)
                    const uint32_t at = atomicAdd(a.bright_n, 1u);
                    if (at < a.bright_cap) a.bright_list[at] = make_uint2(((uint32_t)(f0 + (int)fe) << 16) | (ge * 4u + j), row);
                }
            }
        }
        // (listed by a scan of the lanes' counts, as in k_stream_u16)
        const uint32_t todo_n = (uint32_t)__popc(todo);
        const uint32_t todo_incl = wave_inclusive_scan(todo_n);
        const int T = __builtin_amdgcn_readlane((int)todo_incl, 63);
        {
            uint32_t pos = todo_incl - todo_n, w = todo;
            while (w) {
                const uint32_t j = (uint32_t)__ffs((int)w) - 1u;
                w &= w - 1u;
                s_list[pos++] = (uint16_t)(((uint32_t)lane << 2) | j);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < T; base += 64) {
            const int idx = base + lane;
            if (idx < T) {
                const uint32_t ent = s_list[idx], e = ent >> 2, j = ent & 3u;
                const uint32_t x = s_q[j][e], pv = s_q[4 + j][e];
                const uint32_t m = (s_q[9][e] >> (8 * j)) & 0xFFu;
                unsigned long long y = 0;   // window j = cq[j .. j+6], exact
#pragma unroll
                for (uint32_t t = 0; t < 7; ++t)
                    y += ((unsigned long long)s_q[22 + j + t][e] << 32) | s_q[12 + j + t][e];
                bool certain = false, strong = false;
                if (y < (1ull << 46)) strong = exact_predicate_nosqrt(a, m, x, y, pv, certain);  // m y < 2^53: a exact in float64
                if (!certain) strong = exact_predicate(a, m, x, y, pv);
                if (strong) atomicOr(&s_q[10][e], 1u << j);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (a.wlog) {   // (wave-uniform) wave logs instead of plane, counters and bitmap: see k_stream_u16
            const uint32_t cbm = have ? s_q[10][lane] : 0u;   // strong | undecided << 8
            const uint32_t pxw[4] = {s_q[4][lane], s_q[5][lane], s_q[6][lane], s_q[7][lane]};   // the group's four pixels
            const unsigned long long wm = __builtin_amdgcn_ballot_w64(cbm != 0u);
            if (wm) {
                const uint32_t cnt = (uint32_t)__popcll(wm);
                if (nbuf + cnt > 64u) flush_log();
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u));
                const uint32_t dump = nbuf + cnt < 64u ? nbuf + cnt : 0u;   // (a lane outside the run's new part)
                const int to = (int)((cbm != 0u ? nbuf + rank : dump) << 2);
                const uint32_t e0 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)((row << 16) | ge));
                const uint32_t e1 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)((fe << 16) | cbm));
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[0]), p1 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[1]);
                const uint32_t p2 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[2]), p3 = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)pxw[3]);
                if ((uint32_t)lane - nbuf < cnt) {
                    lbuf[0] = e0; lbuf[1] = e1;
                    lbuf[2] = p0; lbuf[3] = p1; lbuf[4] = p2; lbuf[5] = p3;
                }
                nbuf += cnt;
            }
            qn = 0;
            return;
        }
        if (have) {
            const uint32_t cb = s_q[10][lane];
            if (cb) {  // the plane is all zero when the kernel starts; a lane pair shares a byte, so OR the nibble in
                const uint32_t x0 = ge * 4u;
                uint32_t* plane = reinterpret_cast<uint32_t*>(a.bits + (uint64_t)(f0 + (int)fe) * a.plane_frame_stride + (uint64_t)row * a.mpitch);
                atomicOr(plane + (x0 >> 5), cb << (x0 & 31u));
                atomicAdd(a.tile_counts + (uint64_t)(f0 + (int)fe) * a.n_tiles + (row / (uint32_t)kTileRows), (uint32_t)__popc(cb));
                const uint32_t ob = row * a.occ_spr + (x0 >> 7);
                atomicOr(a.occ + (uint64_t)(f0 + (int)fe) * a.occ_frame_words + (ob >> 5), 1u << (ob & 31u));
            }
        }
        qn = 0;
    };

#pragma unroll
    for (int s = 0; s < KAHEAD; ++s) fetch(s, s);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        mask_row(s, (s + NS - 1) % NS);
        push(s, (s + KAHEAD) % NS);
        fetch((s + KAHEAD) % NS, s + KAHEAD);
    }

    for (int base = 6;; base += NS) {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int s = (6 + t) % NS;
            const int so = (s + KAHEAD) % NS;
            const int sc = (s + NS - 3) % NS;
            const int i = base + t;
            if (i >= total) goto rows_done;
            {
                const uint32_t info = rinfo[s];
                mask_row(s, (s + NS - 1) % NS);
                push(s, so);
                fetch(so, i + KAHEAD);

                // column sums c[-3..6] = L1 L2 L3 c0..c3 R0 R1 R2; every neighbour value rides on an add
                const uint32_t s01 = col[0] + col[1], s23 = col[2] + col[3];
                const uint32_t S = s01 + s23;
                const uint32_t TL = dpp_shr_add(s23, S);       // L2 L3 c0..c3
                const uint32_t TR = dpp_shl_add(s01, S);       // c0..c3 R0 R1
                uint32_t Wn[4];
                Wn[0] = dpp_shr_add(col[1], TL);
                Wn[1] = dpp_shl_add(col[0], TL);
                Wn[2] = dpp_shr_add(col[3], TR);
                Wn[3] = dpp_shl_add(col[2], TR);

                const int yout = __builtin_amdgcn_readfirstlane(yb0 + (i - 6));
                if constexpr (DENSE) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, r_sb, off_byte_st, (uint32_t)yout * a.bpitch, 2);

                const uint32_t xmin = min(min(min(Wn[0], Wn[1]), Wn[2]), Wn[3]);
                const uint32_t pmax = max(max(max(ring[sc][0], ring[sc][1]), ring[sc][2]), ring[sc][3]);
                const uint32_t mmin = (info >> 8) & 0xFFu, mmax = (info >> 16) & 0xFFu;
                const int32_t B = (int32_t)(mmax * pmax) - (int32_t)xmin;   // (garbage when a pixel >= 2^24 is near: flagged below anyway)
                const float bf = (float)B, tf = (float)xmin * (float)mmin;
                const bool pass = bf * __builtin_fabsf(bf) > kS * tf;
                // a valid pixel >= 2^24 in the 7 rows of this lane or of a neighbour
                const uint32_t bl = bighist != 0u ? 1u : 0u;
                bool near_big = false;
                if (__ballot(bl != 0u) != 0ull) near_big = (bl | from_left(bl) | from_right(bl)) != 0u && mmax != 0u;
                const bool want = pass || near_big;
                const unsigned long long fm = FFS_DBG(a, 1) ? 0ull : (__builtin_amdgcn_ballot_w64(want) & owned_mask);
                const bool flag = want && owned;
                if (fm) {
                    const int nfl = __popcll(fm);
                    if (qn + nfl > kQCap) drain();
                    uint32_t ql[4], qh[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) { ql[w] = (uint32_t)colq[w]; qh[w] = (uint32_t)((unsigned long long)colq[w] >> 32); }
                    const uint32_t QL1 = from_left(ql[1]), QL2 = from_left(ql[2]), QL3 = from_left(ql[3]);
                    const uint32_t QR0 = from_right(ql[0]), QR1 = from_right(ql[1]), QR2 = from_right(ql[2]);
                    const uint32_t HL1 = from_left(qh[1]), HL2 = from_left(qh[2]), HL3 = from_left(qh[3]);
                    const uint32_t HR0 = from_right(qh[0]), HR1 = from_right(qh[1]), HR2 = from_right(qh[2]);
                    if (flag && !FFS_DBG(a, 32)) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        const int e = qn + (int)rank;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            s_q[w][e] = Wn[w];
                            s_q[4 + w][e] = ring[sc][w];
                            s_q[15 + w][e] = ql[w];
                            s_q[25 + w][e] = qh[w];
                        }
                        s_q[8][e] = info;   // byte 3: mask bits of the centre row
                        s_q[12][e] = QL1; s_q[13][e] = QL2; s_q[14][e] = QL3;
                        s_q[19][e] = QR0; s_q[20][e] = QR1; s_q[21][e] = QR2;
                        s_q[22][e] = HL1; s_q[23][e] = HL2; s_q[24][e] = HL3;
                        s_q[29][e] = HR0; s_q[30][e] = HR1; s_q[31][e] = HR2;
                        s_q[11][e] = ((uint32_t)yout << 6) | (uint32_t)lane | (near_big ? 0x80000000u : 0u);
                    }
                    qn += nfl;
                }
            }
        }
    }
rows_done:
    if (qn > 0) drain();
    if (a.wlog) {
        flush_log();
        if (lane == 0) a.wlog_n[slot] = nlog;
    }
}

template __global__ void k_bright_fix<uint16_t>(const ThresholdArgs);
template __global__ void k_bright_fix<uint16_t, true>(const ThresholdArgs);
template __global__ void k_bright_fix<uint32_t>(const ThresholdArgs);

template __global__ void k_stream_u16<2, false, false>(const ThresholdArgs);
template __global__ void k_stream_u16<3, false, false>(const ThresholdArgs);
template __global__ void k_stream_u16<4, false, false>(const ThresholdArgs);
template __global__ void k_stream_u16<2, false, true>(const ThresholdArgs);
template __global__ void k_stream_u16<2, true, false>(const ThresholdArgs);
template __global__ void k_stream_u16<2, true, true>(const ThresholdArgs);
template __global__ void k_stream_u32<2, false>(const ThresholdArgs);
template __global__ void k_stream_u32<3, false>(const ThresholdArgs);
template __global__ void k_stream_u32<2, true>(const ThresholdArgs);

}  // namespace ffsamd
