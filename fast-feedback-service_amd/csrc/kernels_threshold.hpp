// kernels_threshold.hpp (included by ffs_api.hip) -- dispersion thresholding as two kernels: a conservative
// streaming candidate kernel and an exact kernel on its candidates.  Since round 2 the default path for 16-bit
// pixels is the ONE-kernel formulation of kernels_stream.hpp; what is here serves 32-bit pixels, the first pass
// of the extended algorithm and the A/B variants (FFS_K1_VARIANT=0/1).
//
// What the reference does: one 7x7 masked window sum per pixel from a shared-memory tile and a
// float32 test (spotfinder/kernels/thresholding.cu:60-125, :145-234).  What "bit-exact" is judged
// against: the float64 summed-area-table predicate of baseline/spotfinder/standalone.cc:113-174.
//
//   K1 `k_candidates_*` streams the frame once.  A wave64 marches down a 512-px column strip with a
//      7-row register ring; the masked pixel value and the valid count share ONE 32-bit word
//      (value + 2^22 per valid pixel: 49*65535 < 2^22, 49 < 2^6), so the exact integer window
//      sums {sum p, n} cost one running vertical add/sub and one sliding horizontal add/sub per
//      pixel; neighbours across lanes come from DPP wave shifts.  Variant 0 (`<false>`): a conservative
//      float32 form of the signal test per pixel, no sum of squares, no LDS.  Variant 1 (`<true>`, and
//      `k_candidates_u32_q`): also a running column sum of p^2, a group screen per lane and row, an LDS
//      queue of the groups that pass and per-pixel conservative signal + dispersion tests on dense lanes
//      when the queue drains -- the candidate plane then holds little more than the true strong pixels.
//      Both emit a 1-bit/pixel candidate plane (a superset of the strong pixels) and zero-fill the byte mask.
//   K2 `k_exact*`      visits only the candidates: exact integer 7x7 sums {n, sum p, sum p^2}
//      and the oracle's fp64 predicate, operation for operation; clears failed candidates in the
//      bit plane (it becomes the strong plane), sets the byte mask, counts the strong pixels per tile.
#pragma once
#include "ffs_device.h"

namespace ffsamd {

// ================================================================================================
// K1: candidates, uint16 pixels
// ================================================================================================

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    // raw buffer, stride 0; DST_SEL/format word as in the CDNA guides (0x00020000)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

constexpr uint32_t kFlag = 1u << 22;       // one valid pixel
constexpr uint32_t kXMask = kFlag - 1u;    // low 22 bits: sum of pixel values

// lane i <- lane i-1 (lane 0 gets 0)
__device__ __forceinline__ uint32_t from_left(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
// lane i <- lane i+1 (lane 63 gets 0)
__device__ __forceinline__ uint32_t from_right(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}

struct RowRegsU16 {
    uint4 raw;     // 8 pixels
    uint32_t mb;   // 8 valid bits
};

__device__ __forceinline__ void unpack_u16(const RowRegsU16& r, uint32_t (&A)[8]) {
    const uint32_t w[4] = {r.raw.x, r.raw.y, r.raw.z, r.raw.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        A[2 * q] = (w[q] & 0xFFFFu) | kFlag;
        A[2 * q + 1] = __builtin_amdgcn_alignbit(kFlag >> 16, w[q], 16);  // (w >> 16) | 2^22
    }
    if (__ballot(r.mb != 0xFFu) != 0ull) {
#pragma unroll
        for (int j = 0; j < 8; ++j) A[j] &= (uint32_t)__builtin_amdgcn_sbfe((int)r.mb, j, 1);  // 0 / ~0
    }
}

// The per-pixel conservative signal test (see k_candidates_u16) on one lane-group of 8 pixels:
// window words Wn[j] (sum p | count << 22) and centre words Ac[j]; returns the 8 candidate bits.
//   oracle: b = m p - x > nsig_s sqrt(x m)   (standalone.cc:167,169-170)
//   here:   b |b| > nsig_s^2 (1 - 2^-16) x m  in float32 (b exact, |b| < 2^22).
// An invalid centre has A = 0 -> p = 0 -> b <= 0 -> never a candidate; m < 2 gives b = 0 likewise.
// e = kS x m - b|b| is negative exactly for candidates; its sign bit is shifted into the byte
// with one v_alignbit per pixel (j = 7 first).
__device__ __forceinline__ uint32_t signal_test8(const uint32_t (&Wn)[8], const uint32_t (&Ac)[8], float kS) {
    uint32_t cb = 0;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const uint32_t x = Wn[j] & kXMask;
        const uint32_t m = Wn[j] >> 22;
        const uint32_t pv = Ac[j] & kXMask;
        const int32_t b = (int32_t)(m * pv) - (int32_t)x;  // 24-bit multiplies
        const uint32_t tq = x * m;
        const float bf = (float)b;
        const float tf = (float)tq;
        const float lhs = bf * __builtin_fabsf(bf);
        const float e = __builtin_fmaf(kS, tf, -lhs);
        cb = __builtin_amdgcn_alignbit(cb, __float_as_uint(e), 31);
    }
    return cb;
}


constexpr int kQCap = 64;        // lane-group queue entries per wave (8 KB of LDS)

// Per-pixel tests on one queued lane-group (8 pixels): the conservative signal test above AND a
// conservative form of the oracle's dispersion test
//     a = m y - x^2 - x (m-1)  >  c = nsig_b x sqrt(2 (m-1))        (standalone.cc:166,168,170)
// in float32 with an explicit error allowance (|fl(a) - a| < 2^-21 m y; we grant 2^-20 m y and
// shave 2^-20 off c).  y = sum p^2 comes from 32-bit running sums that are exact while every
// pixel of the window is < 8192 (49 * 8191^2 < 2^32); x < 8192 guarantees that, and brighter
// windows are passed on unconditionally (the exact kernel decides them).
// The queue lives in LDS word-major, q[w][e]: words 0-7 window words W, 8-15 centre words A,
// 16-29 the 14 column sums of p^2 (L5 L6 L7 c0..c7 R0 R1 R2), 30 the (row, lane) tag.  Word-major
// keeps every access conflict-free (consecutive lanes -> consecutive entries) and lets the pushes
// be ds_write2_b32 from whatever registers hold the values (a b128 layout costs ~30 v_mov per push).
constexpr int kQWords = 32;
// Deliberately a rolled loop reading LDS word by word: the drain runs once per ~20 rows, and
// keeping its live registers to a handful is what lets the streaming loop keep 4 waves per SIMD.
// EXT (extended algorithm's first pass): the dispersion test alone decides.
// EXT also classifies its positives: bit j of `sure` is set when pixel j passes the dispersion test with
// room to spare (a - 2^-20 m y > c (1 + 2^-18): float32 rounding and the shaved nsig_b cannot turn that
// around), its centre is valid, the window holds at least min_count pixels and the 32-bit sum of p^2 is
// exact (x < 8192).  Those pixels need no second opinion from the exact kernel.
template <bool EXT>
__device__ __forceinline__ uint32_t group_tests8(const uint32_t (*q)[kQCap], int e, float kS, float kB,
                                                 uint32_t min_count = 2, uint32_t* sure = nullptr) {
    uint32_t wq = 0;  // window j sums cq[j .. j+6]
    uint32_t sb = 0;
#pragma nounroll
    for (int t = 0; t < 7; ++t) wq += q[16 + t][e];
    uint32_t cb = 0;
#pragma unroll 2
    for (int j = 0; j < 8; ++j) {
        const uint32_t W = q[j][e], A = q[8 + j][e];
        const uint32_t x = W & kXMask, m = W >> 22, pv = A & kXMask;
        const int32_t b = (int32_t)(m * pv) - (int32_t)x;
        const float bf = (float)b, tf = (float)(x * m);
        const bool sig = EXT || bf * __builtin_fabsf(bf) > kS * tf;
        const float mf = (float)m, xf = (float)x, yf = (float)wq;
        const float t0 = mf * yf;
        const float af = (t0 - xf * xf) - xf * (mf - 1.0f);
        const float cf = xf * (kB * __builtin_amdgcn_sqrtf(2.0f * (mf - 1.0f)));  // raw v_sqrt_f32: 1 ulp, inside the 2^-20 allowances
        const bool disp = (af + t0 * 9.5367431640625e-07f >= cf) || x >= 8192u;
        cb |= (sig && disp) ? (1u << j) : 0u;
        if constexpr (EXT) {
            const bool certain = (af - t0 * 9.5367431640625e-07f > cf * (1.0f + 3.814697265625e-06f)) && x < 8192u
                                 && m >= min_count && (A >> 22) != 0u;
            sb |= certain ? (1u << j) : 0u;
        }
        wq = wq - q[16 + j][e] + q[23 + j][e];  // j = 7 reads the tag word; that sum is not used
    }
    if (sure) *sure = sb;
    return cb;
}

// SCREEN = false: every pixel takes the conservative signal test (11 VALU ops / pixel); about
//   0.5-0.7 % of pixels become candidates for the exact kernel.
// SCREEN = true : the kernel also carries sum p^2 (a second running column sum, updated with
//   dq = (p_in - p_out)(p_in + p_out)), screens each lane's 8-pixel group as a whole with the signal
//   test -- largest centre pixel against smallest window sum, still conservative -- and queues the
//   few groups that pass (a few %) in LDS with their window words; 64 queued groups at a time take
//   the per-pixel signal AND dispersion tests on dense lanes (group_tests8).  The candidate plane
//   then holds little more than the true strong pixels, so the exact kernel no longer re-reads the
//   batch from HBM.  Both variants are supersets of the oracle's strong pixels.
// EXT = true (needs SCREEN): first pass of the extended algorithm (baseline.cpp:415-475).  The plane
//   written is a.dplane and holds candidates for "index of dispersion above background": the group
//   screen bounds a = m y - x^2 - x (m-1) from above with the largest sum p^2 and the smallest sum p
//   of the group's eight windows, queued groups take the per-pixel dispersion test.
template <bool SCREEN, bool EXT = false>
__global__ __launch_bounds__(64, 4) void k_candidates_u16(const ThresholdArgs a) {  // <= 128 VGPRs: 4 waves per SIMD
    static_assert(SCREEN || !EXT, "the extended first pass is built on the screening variant");
    __shared__ uint32_t s_q[SCREEN ? kQWords : 1][SCREEN ? kQCap : 1];

    const int lane = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group), and
    // neighbouring strips of one band share the cache lines at their common edge: give all strips
    // of a band the same label so those lines are served by one L2.  Speed only, never correctness.
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int strip = q % a.n_strips;
    const int band = xcd + 8 * (q / a.n_strips);
    if (band >= a.n_bands) return;  // grid is padded to a multiple of 8 bands
    const int frame = blockIdx.y;
    const int yb0 = band * a.band_rows;
    const int yb1 = min(yb0 + a.band_rows, a.H);
    const int sx0 = strip * kStripOwnedPx + kStripStartOffset;  // multiple of 8
    const int lx0 = sx0 + lane * kLanePx;
    const bool active = lx0 >= 0 && lx0 + kLanePx <= a.pitch_px;
    const bool owned = active && lane >= 1 && lane <= 62;
    const int cx = active ? lx0 : 0;

    // Buffer resources (wave-uniform, SGPRs): per-lane column offset in a VGPR, row offset in an
    // SGPR, so no per-row VALU address arithmetic; out-of-range accesses are dropped by hardware.
    const rsrc_t r_img = make_rsrc((const uint8_t*)a.image + (uint64_t)frame * a.frame_stride,
                                   (uint32_t)a.H * a.pitch);
    const rsrc_t r_mask = make_rsrc(a.maskbits, (uint32_t)a.H * a.mpitch);
    const rsrc_t r_sb = make_rsrc(a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride,
                                  (uint32_t)a.H * a.bpitch);
    const rsrc_t r_cb = make_rsrc((EXT ? a.dplane : a.bits) + (uint64_t)frame * a.plane_frame_stride,
                                  (uint32_t)a.H * a.mpitch);
    // EXT: a.bits receives the positives that still need the exact kernel ("uncertain"); a.dplane all of them
    const rsrc_t r_ub = make_rsrc(a.bits + (uint64_t)frame * a.plane_frame_stride, (uint32_t)a.H * a.mpitch);
    // Offsets with bit 31 set are out of range for every resource: such loads return 0 and such
    // stores are dropped.  Used instead of branches (inactive / not-owned lanes, rows outside the
    // image) so that the loop body is straight-line code and the compiler can keep several rows of
    // loads in flight with exact s_waitcnt vmcnt(N) counts.
    constexpr uint32_t kOob = 0x80000000u;
    const uint32_t off_px = (uint32_t)cx * 2u;
    const uint32_t off_bit = active ? ((uint32_t)cx >> 3) : kOob;          // mask-bit loads
    // the byte mask is zero-filled in aligned 512-byte runs (full cache lines), independent of
    // which lanes own which pixels: wave (strip s) clears columns [512 s, 512 s + 512)
    const uint32_t zcol = (uint32_t)strip * 512u + (uint32_t)lane * 8u;
    const uint32_t off_byte_st = zcol < (uint32_t)a.bpitch ? zcol : kOob;  // byte-mask stores
    const uint32_t off_bit_st = owned ? ((uint32_t)cx >> 3) : kOob;        // candidate stores

    const int total = (yb1 - yb0) + 6;  // incoming rows yb0-3 .. yb1+2
    const float kS = a.kS, kB = a.kB;

    uint32_t ring[7][8];
    uint32_t col[8], colq[8];
    RowRegsU16 pre[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ring[s][j] = 0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { col[j] = 0; colq[j] = 0; }

    // issue the loads of incoming row number i (image row yb0 - 3 + i)
    auto fetch = [&](RowRegsU16& dst, int i) {
        const int yin = yb0 - 3 + i;
        const bool ok = (i < total) & (yin >= 0) & (yin < a.H);  // wave-uniform (scalar ALU)
        const uint32_t kill = ok ? 0u : kOob;
        const uint32_t row = ok ? (uint32_t)yin : 0u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_img, off_px | kill, row * a.pitch, 0);
        dst.raw = make_uint4(v[0], v[1], v[2], v[3]);
        dst.mb = __builtin_amdgcn_raw_buffer_load_b8(r_mask, off_bit | kill, row * a.mpitch, 0);
    };

    // vertical running sums: add the incoming row, retire the row that left the 7-row window
    auto push = [&](int s, const uint32_t (&A)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SCREEN) {
                // p_in^2 - p_out^2 as one 24-bit multiply; an invalid pixel has low half 0
                const int32_t pn = (int32_t)(A[j] & 0xFFFFu), po = (int32_t)(ring[s][j] & 0xFFFFu);
                colq[j] += (uint32_t)((pn - po) * (pn + po));
            }
            col[j] += A[j] - ring[s][j];
            ring[s][j] = A[j];
        }
    };

    int qn = 0;  // queued lane-groups (wave-uniform)
    auto drain = [&]() {
        if constexpr (SCREEN) if (lane < qn) {
            uint32_t sure = 0;
            const uint32_t cb = group_tests8<EXT>(s_q, lane, kS, kB, (uint32_t)a.min_count, &sure);
            const uint32_t tag = s_q[30][lane], row = tag >> 6, ln = tag & 63u;
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)cb, r_cb, row * a.mpitch + (uint32_t)(sx0 >> 3) + ln, 0, 0);
            if constexpr (EXT) {
                const uint32_t uncertain = a.max_valid >= 0 ? cb : (cb & ~sure);  // a trusted-range test needs the pixel
                if (uncertain)
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)uncertain, r_ub, row * a.mpitch + (uint32_t)(sx0 >> 3) + ln, 0, 0);
            }
        }
        qn = 0;
    };

    constexpr int kAhead = SCREEN ? 2 : 4;  // rows of loads in flight per wave (measured: 1, 2, 3 within 2 %)
#pragma unroll
    for (int s = 0; s < kAhead; ++s) fetch(pre[s], s);

    // warm-up: rows 0..5 of the band's input only fill the window
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        uint32_t A[8];
        unpack_u16(pre[s], A);
        fetch(pre[(s + kAhead) % 7], s + kAhead);
        push(s, A);
    }

    for (int base = 6;; base += 7) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int s = (6 + t) % 7;   // slot of incoming row i (i % 7 == s)
            const int sc = (s + 4) % 7;  // slot of the centre row i - 3
            const int i = base + t;
            if (i >= total) goto rows_done;
            {
                uint32_t A[8];
                unpack_u16(pre[s], A);
                fetch(pre[(s + kAhead) % 7], i + kAhead);
                push(s, A);

                // horizontal 7-tap over column sums c[-3..10] = L5 L6 L7 c0..c7 R0 R1 R2
                const uint32_t L5 = from_left(col[5]), L6 = from_left(col[6]), L7 = from_left(col[7]);
                const uint32_t R0 = from_right(col[0]), R1 = from_right(col[1]), R2 = from_right(col[2]);
                uint32_t Wn[8];
                if constexpr (!SCREEN) {
                    // two independent sliding chains (outwards from pixels 3 and 4) for ILP
                    const uint32_t mid = (col[1] + col[2] + col[3]) + (col[4] + col[5] + col[6]);  // c1..c6
                    Wn[3] = mid + col[0];
                    Wn[4] = mid + col[7];
                    Wn[2] = Wn[3] - col[6] + L7;
                    Wn[5] = Wn[4] - col[1] + R0;
                    Wn[1] = Wn[2] - col[5] + L6;
                    Wn[6] = Wn[5] - col[2] + R1;
                    Wn[0] = Wn[1] - col[4] + L5;
                    Wn[7] = Wn[6] - col[3] + R2;
                } else {
                    // one chain: fewer values live at once (this variant is register-bound)
                    Wn[0] = (L5 + L6 + L7) + (col[0] + col[1] + col[2]) + col[3];
                    Wn[1] = Wn[0] - L5 + col[4];
                    Wn[2] = Wn[1] - L6 + col[5];
                    Wn[3] = Wn[2] - L7 + col[6];
                    Wn[4] = Wn[3] - col[0] + col[7];
                    Wn[5] = Wn[4] - col[1] + R0;
                    Wn[6] = Wn[5] - col[2] + R1;
                    Wn[7] = Wn[6] - col[3] + R2;
                }

                // readfirstlane: the row offsets are wave-uniform, but the compiler keeps the loop's
                // induction value in a VGPR and would wrap every store in a waterfall loop
                const int yout = __builtin_amdgcn_readfirstlane(yb0 + (i - 6));
                const uint32_t so_bytes = (uint32_t)yout * a.bpitch, so_bits = (uint32_t)yout * a.mpitch;
                if constexpr (!SCREEN) {
                    const uint32_t cb = signal_test8(Wn, ring[sc], kS);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, r_sb, off_byte_st, so_bytes, 0);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)cb, r_cb, off_bit_st, so_bits, 0);
                } else {
                    // group screen: every pixel j of the group has p_j <= pmax and, when all eight
                    // windows hold the same count m, x_j >= xmin; b_j <= m pmax - xmin and
                    // sqrt(x_j m) >= sqrt(xmin m), so a group whose (pmax, xmin) fails the test has
                    // no candidate.  Groups with unequal counts (next to masked pixels) always pass.
                    const uint32_t wmin = min(min(min(Wn[0], Wn[1]), min(Wn[2], Wn[3])),
                                              min(min(Wn[4], Wn[5]), min(Wn[6], Wn[7])));
                    auto window_max = [&]() {
                        return max(max(max(Wn[0], Wn[1]), max(Wn[2], Wn[3])), max(max(Wn[4], Wn[5]), max(Wn[6], Wn[7])));
                    };
                    const uint32_t x = wmin & kXMask, m = wmin >> 22;
                    bool pass;
                    uint32_t QL5 = 0, QL6 = 0, QL7 = 0, QR0 = 0, QR1 = 0, QR2 = 0;
                    if constexpr (EXT) {
                        // the group's 14 column sums of p^2 (own 8 + 3 from each neighbour lane); the
                        // eight window sums slide over them, only their maximum is kept
                        QL5 = from_left(colq[5]); QL6 = from_left(colq[6]); QL7 = from_left(colq[7]);
                        QR0 = from_right(colq[0]); QR1 = from_right(colq[1]); QR2 = from_right(colq[2]);
                        uint32_t wq = (QL5 + QL6 + QL7) + (colq[0] + colq[1] + colq[2]) + colq[3];
                        uint32_t ymax = wq;
                        wq += colq[4] - QL5; ymax = max(ymax, wq);
                        wq += colq[5] - QL6; ymax = max(ymax, wq);
                        wq += colq[6] - QL7; ymax = max(ymax, wq);
                        wq += colq[7] - colq[0]; ymax = max(ymax, wq);
                        wq += QR0 - colq[1]; ymax = max(ymax, wq);
                        wq += QR1 - colq[2]; ymax = max(ymax, wq);
                        wq += QR2 - colq[3]; ymax = max(ymax, wq);
                        // a_j <= m ymax - xmin^2 - xmin (m-1) and c_j >= nsig_b xmin sqrt(2 (m-1)) for
                        // every pixel j of a group whose windows hold the same count m; float32 with
                        // the same allowance as group_tests8.  The 32-bit sums of p^2 are exact while
                        // x < 8192; brighter groups, and groups with unequal counts, always pass.
                        const float mf = (float)m, xf = (float)x, yf = (float)ymax;
                        const float t0 = mf * yf;
                        const float af = (t0 - xf * xf) - xf * (mf - 1.0f);
                        const float cf = xf * (kB * __builtin_amdgcn_sqrtf(2.0f * (mf - 1.0f)));  // raw v_sqrt_f32: 1 ulp, inside the 2^-20 allowances
                        const uint32_t wmax = window_max();
                        pass = (af + t0 * 9.5367431640625e-07f >= cf) || (wmax & kXMask) >= 4096u
                               || ((wmin ^ wmax) >> 22) != 0;
                    } else {
                        const uint32_t amax = max(max(max(ring[sc][0], ring[sc][1]), max(ring[sc][2], ring[sc][3])),
                                                  max(max(ring[sc][4], ring[sc][5]), max(ring[sc][6], ring[sc][7])));
                        const uint32_t pv = amax & kXMask;
                        const int32_t b = (int32_t)(m * pv) - (int32_t)x;
                        const float bf = (float)b, tf = (float)(x * m);
                        // a smallest count of 49 means all eight windows are full: counts can differ only
                        // where some lane's minimum is below 49 (wave-uniform test; inside a detector
                        // module the window maximum is never computed)
                        bool unequal = false;
                        if (__ballot(m != 49u) != 0ull) unequal = ((wmin ^ window_max()) >> 22) != 0;
                        pass = (bf * __builtin_fabsf(bf) > kS * tf) || unequal;
                    }
                    const bool flag = owned && pass;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, r_sb, off_byte_st, so_bytes, 0);
                    // queued groups get their byte from drain(); everybody else stores 0 now
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, r_cb, flag ? kOob : off_bit_st, so_bits, 0);
                    // (EXT: the plane of uncertain positives is zeroed by the host before the launch; only the
                    // drain writes into it)
                    const unsigned long long fm = __ballot(flag);
                    if (fm) {  // wave-uniform
                        const int nf = __popcll(fm);
                        if (qn + nf > kQCap) drain();
                        if constexpr (!EXT) {
                            // the group's 14 column sums of p^2 (own 8 + 3 from each neighbour lane)
                            QL5 = from_left(colq[5]); QL6 = from_left(colq[6]); QL7 = from_left(colq[7]);
                            QR0 = from_right(colq[0]); QR1 = from_right(colq[1]); QR2 = from_right(colq[2]);
                        }
                        if (flag) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                            const int e = qn + (int)rank;
                            if (e >= kQCap) {
                                // queue full (a burst of flagged groups): hand the whole group to
                                // the exact kernel instead -- still a superset, never a miss
                                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0xFF, r_cb, off_bit_st, so_bits, 0);
                                if constexpr (EXT)
                                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0xFF, r_ub, off_bit_st, so_bits, 0);
                            } else {
#pragma unroll
                            for (int w = 0; w < 8; ++w) {
                                s_q[w][e] = Wn[w];
                                s_q[8 + w][e] = ring[sc][w];
                                s_q[19 + w][e] = colq[w];
                            }
                            s_q[16][e] = QL5; s_q[17][e] = QL6; s_q[18][e] = QL7;
                            s_q[27][e] = QR0; s_q[28][e] = QR1; s_q[29][e] = QR2;
                            s_q[30][e] = ((uint32_t)yout << 6) | (uint32_t)lane;
                            }
                        }
                        qn = min(qn + nf, kQCap);
                    }
                }
            }
        }
    }
rows_done:
    if constexpr (SCREEN) {
        if (qn > 0) drain();
    }
}
template __global__ void k_candidates_u16<false>(const ThresholdArgs);
template __global__ void k_candidates_u16<true>(const ThresholdArgs);
template __global__ void k_candidates_u16<true, true>(const ThresholdArgs);

// ================================================================================================
// K1: candidates, uint32 pixels (the reference's PIXEL_DATA_32BIT build, h5read.h:16-20)
// ================================================================================================
// Same structure with 4 pixels (16 B) per lane.  The oracle only sums pixels < 2^24
// (standalone.cc:78,90), so sum p < 2^30 and the count no longer shares a word with it: two
// running words per pixel (X = sum p, M = count).  Lanes 0,1 and 62,63 are halo; lanes 2..61 own
// 240 px, so that an (even, odd) lane pair owns exactly one byte of the candidate plane.
constexpr int kLanePx32 = 4;
constexpr int kStripOwnedPx32 = 60 * kLanePx32;  // 240
constexpr int kStripStartOffset32 = -8;

struct RowRegsU32 {
    uint4 raw;     // 4 pixels
    uint32_t mb;   // 4 valid bits
};

__global__ __launch_bounds__(64) void k_candidates_u32(const ThresholdArgs a) {
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;  // XCD-aware mapping, see k_candidates_u16
    const int strip = q % a.n_strips;
    const int band = xcd + 8 * (q / a.n_strips);
    if (band >= a.n_bands) return;
    const int frame = blockIdx.y;
    const int yb0 = band * a.band_rows;
    const int yb1 = min(yb0 + a.band_rows, a.H);
    const int lx0 = strip * kStripOwnedPx32 + kStripStartOffset32 + lane * kLanePx32;
    const bool active = lx0 >= 0 && lx0 + kLanePx32 <= a.pitch_px;
    const bool owned = active && lane >= 2 && lane <= 61;
    const int cx = active ? lx0 : 0;

    const rsrc_t r_img = make_rsrc((const uint8_t*)a.image + (uint64_t)frame * a.frame_stride,
                                   (uint32_t)a.H * a.pitch);
    const rsrc_t r_mask = make_rsrc(a.maskbits, (uint32_t)a.H * a.mpitch);
    const rsrc_t r_sb = make_rsrc(a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride,
                                  (uint32_t)a.H * a.bpitch);
    const rsrc_t r_cb = make_rsrc(a.bits + (uint64_t)frame * a.plane_frame_stride,
                                  (uint32_t)a.H * a.mpitch);
    constexpr uint32_t kOob = 0x80000000u;  // out-of-range offset: loads give 0, stores are dropped
    const uint32_t off_px = (uint32_t)cx * 4u;
    const uint32_t off_bit = active ? ((uint32_t)cx >> 3) : kOob;
    const uint32_t zcol = (uint32_t)strip * 256u + (uint32_t)lane * 4u;  // aligned 256-byte zero runs
    const uint32_t off_byte_st = zcol < (uint32_t)a.bpitch ? zcol : kOob;
    const uint32_t off_bit_st = (owned && !(lane & 1)) ? ((uint32_t)cx >> 3) : kOob;
    const uint32_t nib = (uint32_t)cx & 4u;

    const int total = (yb1 - yb0) + 6;
    const float kS = a.kS;

    uint32_t ringX[7][4], ringF[7];  // masked pixel values; flags: bits 0-3 summed, bits 4-7 mask
    uint32_t colX[4], colM[4];
    RowRegsU32 pre[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        ringF[s] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) ringX[s][j] = 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { colX[j] = 0; colM[j] = 0; }

    auto fetch = [&](RowRegsU32& dst, int i) {
        const int yin = yb0 - 3 + i;
        const bool ok = (i < total) & (yin >= 0) & (yin < a.H);  // wave-uniform
        const uint32_t kill = ok ? 0u : kOob;
        const uint32_t row = ok ? (uint32_t)yin : 0u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_img, off_px | kill, row * a.pitch, 0);
        dst.raw = make_uint4(v[0], v[1], v[2], v[3]);
        const uint32_t mb = __builtin_amdgcn_raw_buffer_load_b8(r_mask, off_bit | kill, row * a.mpitch, 0);
        dst.mb = (mb >> nib) & 0xFu;
    };

    auto push = [&](int s, const RowRegsU32& r) {
        const uint32_t p[4] = {r.raw.x, r.raw.y, r.raw.z, r.raw.w};
        uint32_t fl = r.mb << 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool sm = ((r.mb >> j) & 1u) && p[j] < (1u << 24);  // mm, standalone.cc:90
            const uint32_t X = sm ? p[j] : 0u;
            const uint32_t Mn = sm ? 1u : 0u;
            const uint32_t Mo = (ringF[s] >> j) & 1u;
            colX[j] += X - ringX[s][j];
            colM[j] += Mn - Mo;
            ringX[s][j] = X;
            fl |= Mn << j;
        }
        ringF[s] = fl;
    };

#pragma unroll
    for (int s = 0; s < 4; ++s) fetch(pre[s], s);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const RowRegsU32 r = pre[s];
        fetch(pre[(s + 4) % 7], s + 4);
        push(s, r);
    }

    for (int base = 6;; base += 7) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int s = (6 + t) % 7;
            const int sc = (s + 4) % 7;
            const int i = base + t;
            if (i >= total) return;
            {
                const RowRegsU32 r = pre[s];
                fetch(pre[(s + 4) % 7], i + 4);
                push(s, r);

                // c[-3..6] = L1 L2 L3 c0..c3 R0 R1 R2 for both words
                uint32_t WX[4], WM[4];
                {
                    const uint32_t L1 = from_left(colX[1]), L2 = from_left(colX[2]), L3 = from_left(colX[3]);
                    const uint32_t R0 = from_right(colX[0]), R1 = from_right(colX[1]), R2 = from_right(colX[2]);
                    WX[0] = (L1 + L2 + L3) + (colX[0] + colX[1] + colX[2]) + colX[3];
                    WX[1] = WX[0] - L1 + R0;
                    WX[2] = WX[1] - L2 + R1;
                    WX[3] = WX[2] - L3 + R2;
                }
                {
                    const uint32_t L1 = from_left(colM[1]), L2 = from_left(colM[2]), L3 = from_left(colM[3]);
                    const uint32_t R0 = from_right(colM[0]), R1 = from_right(colM[1]), R2 = from_right(colM[2]);
                    WM[0] = (L1 + L2 + L3) + (colM[0] + colM[1] + colM[2]) + colM[3];
                    WM[1] = WM[0] - L1 + R0;
                    WM[2] = WM[1] - L2 + R1;
                    WM[3] = WM[2] - L3 + R2;
                }

                // Conservative signal test.  b = m p - x is evaluated in float32 with an explicit
                // error bound E = 2^-22 (m p + x) added (|fl(b) - b| < E); a valid centre pixel
                // >= 2^24 (outside the sums, but a legal centre) is passed on unconditionally.
                uint32_t cb = 0;
                const uint32_t fc = ringF[sc];
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const float xf = (float)WX[j];
                    const float mf = (float)WM[j];
                    const float pf = (float)ringX[sc][j];
                    const float sp = __builtin_fmaf(mf, pf, xf);
                    const float bf = __builtin_fmaf(mf, pf, -xf);
                    const float u = __builtin_fmaf(sp, 2.384185791015625e-07f, bf);
                    const float lhs = u * __builtin_fabsf(u);
                    const float tf = xf * mf;
                    const float e = __builtin_fmaf(kS, tf, -lhs);
                    const uint32_t big = ((fc >> (4 + j)) & ~(fc >> j)) & 1u;  // masked-in but not summed
                    cb = (cb << 1) | ((__float_as_uint(e) >> 31) | big);
                }

                // an (even, odd) lane pair shares one byte of the candidate plane
                const uint32_t hi = from_right(cb);
                const int yout = __builtin_amdgcn_readfirstlane(yb0 + (i - 6));  // see k_candidates_u16
                __builtin_amdgcn_raw_buffer_store_b32(0u, r_sb, off_byte_st, (uint32_t)yout * a.bpitch, 0);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(cb | (hi << 4)), r_cb, off_bit_st,
                                                     (uint32_t)yout * a.mpitch, 0);
            }
        }
    }
}

// ---- 32-bit pixels with the group screen and the LDS queue (variant 1) ------------------------------
// k_candidates_u32 above tests every pixel for signal only, and ~0.5 % of all pixels reach the exact
// kernel, which then gathers seven rows for each of them (260 us per 32 Jungfrau frames).  This variant
// is the 16-bit kernel's scheme on the 4-pixel lane groups: running column sums of p^2 (pixels clamped
// to 8191 first -- a window that holds a larger pixel has sum p >= 8192 and skips the dispersion test
// anyway --, so the differences fit a 24-bit multiply and the 32-bit sums are exact whenever they are
// used), a group screen (largest centre pixel against smallest window sum), and per-pixel signal AND
// dispersion tests on queued groups.  A lane pair shares one byte of the candidate plane, so the plane
// is zeroed before the launch and the drain ORs the few non-zero nibbles in with atomics.
constexpr int kQWords32 = 24;  // 0-3 WX, 4-7 WM, 8-11 centre X, 12 centre flags, 13-22 column sums of p^2, 23 tag

__device__ __forceinline__ uint32_t group_tests4(const uint32_t (*q)[kQCap], int e, float kS, float kB) {
    uint32_t wq = 0;  // window j sums cq[j .. j+6] of L1 L2 L3 c0 c1 c2 c3 R0 R1 R2
#pragma nounroll
    for (int t = 0; t < 7; ++t) wq += q[13 + t][e];
    const uint32_t fc = q[12][e];
    uint32_t cb = 0;
#pragma unroll 2
    for (int j = 0; j < 4; ++j) {
        const uint32_t x = q[j][e], m = q[4 + j][e], p = q[8 + j][e];
        const float xf = (float)x, mf = (float)m, pf = (float)p;
        // signal, as in k_candidates_u32: b = m p - x in float32 with the error bound 2^-22 (m p + x) added
        const float sp = __builtin_fmaf(mf, pf, xf);
        const float bf = __builtin_fmaf(mf, pf, -xf);
        const float u = __builtin_fmaf(sp, 2.384185791015625e-07f, bf);
        const bool sig = u * __builtin_fabsf(u) > kS * (xf * mf);
        // dispersion, as in group_tests8 (exact sums while x < 8192)
        const float yf = (float)wq;
        const float t0 = mf * yf;
        const float af = (t0 - xf * xf) - xf * (mf - 1.0f);
        const float cf = xf * (kB * __builtin_amdgcn_sqrtf(2.0f * (mf - 1.0f)));
        const bool disp = (af + t0 * 9.5367431640625e-07f >= cf) || x >= 8192u;
        const uint32_t big = ((fc >> (4 + j)) & ~(fc >> j)) & 1u;  // valid centre >= 2^24: the exact kernel decides
        cb |= ((sig && disp) || big) ? (1u << j) : 0u;
        wq = wq - q[13 + j][e] + q[20 + j][e];  // j = 3 reads the tag word; that sum is not used
    }
    return cb;
}

__global__ __launch_bounds__(64) void k_candidates_u32_q(const ThresholdArgs a) {
    __shared__ uint32_t s_q[kQWords32][kQCap];
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int strip = qb % a.n_strips;
    const int band = xcd + 8 * (qb / a.n_strips);
    if (band >= a.n_bands) return;
    const int frame = blockIdx.y;
    const int yb0 = band * a.band_rows;
    const int yb1 = min(yb0 + a.band_rows, a.H);
    const int lx0 = strip * kStripOwnedPx32 + kStripStartOffset32 + lane * kLanePx32;
    const bool active = lx0 >= 0 && lx0 + kLanePx32 <= a.pitch_px;
    const bool owned = active && lane >= 2 && lane <= 61;
    const int cx = active ? lx0 : 0;

    const rsrc_t r_img = make_rsrc((const uint8_t*)a.image + (uint64_t)frame * a.frame_stride, (uint32_t)a.H * a.pitch);
    const rsrc_t r_mask = make_rsrc(a.maskbits, (uint32_t)a.H * a.mpitch);
    const rsrc_t r_sb = make_rsrc(a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride, (uint32_t)a.H * a.bpitch);
    uint32_t* const plane = reinterpret_cast<uint32_t*>(a.bits + (uint64_t)frame * a.plane_frame_stride);  // zeroed by the host
    constexpr uint32_t kOob = 0x80000000u;
    const uint32_t off_px = (uint32_t)cx * 4u;
    const uint32_t off_bit = active ? ((uint32_t)cx >> 3) : kOob;
    const uint32_t zcol = (uint32_t)strip * 256u + (uint32_t)lane * 4u;
    const uint32_t off_byte_st = zcol < (uint32_t)a.bpitch ? zcol : kOob;
    const uint32_t nib = (uint32_t)cx & 4u;
    const int sx0 = strip * kStripOwnedPx32 + kStripStartOffset32;

    const int total = (yb1 - yb0) + 6;
    const float kS = a.kS, kB = a.kB;

    uint32_t ringX[7][4], ringF[7];
    uint32_t colX[4], colM[4], colQ[4];
    RowRegsU32 pre[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        ringF[s] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) ringX[s][j] = 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { colX[j] = 0; colM[j] = 0; colQ[j] = 0; }

    auto fetch = [&](RowRegsU32& dst, int i) {
        const int yin = yb0 - 3 + i;
        const bool ok = (i < total) & (yin >= 0) & (yin < a.H);
        const uint32_t kill = ok ? 0u : kOob;
        const uint32_t row = ok ? (uint32_t)yin : 0u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_img, off_px | kill, row * a.pitch, 0);
        dst.raw = make_uint4(v[0], v[1], v[2], v[3]);
        const uint32_t mb = __builtin_amdgcn_raw_buffer_load_b8(r_mask, off_bit | kill, row * a.mpitch, 0);
        dst.mb = (mb >> nib) & 0xFu;
    };
    auto push = [&](int s, const RowRegsU32& r) {
        const uint32_t p[4] = {r.raw.x, r.raw.y, r.raw.z, r.raw.w};
        uint32_t fl = r.mb << 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool sm = ((r.mb >> j) & 1u) && p[j] < (1u << 24);  // mm, standalone.cc:90
            const uint32_t X = sm ? p[j] : 0u;
            const uint32_t Mn = sm ? 1u : 0u;
            const uint32_t Mo = (ringF[s] >> j) & 1u;
            const int32_t cn = (int32_t)min(X, 8191u), co = (int32_t)min(ringX[s][j], 8191u);
            colQ[j] += (uint32_t)((cn - co) * (cn + co));
            colX[j] += X - ringX[s][j];
            colM[j] += Mn - Mo;
            ringX[s][j] = X;
            fl |= Mn << j;
        }
        ringF[s] = fl;
    };

    int qn = 0;
    auto drain = [&]() {
        if (lane < qn) {
            const uint32_t cb = group_tests4(s_q, lane, kS, kB);
            if (cb) {
                const uint32_t tag = s_q[23][lane], row = tag >> 6, ln = tag & 63u;
                const uint32_t x0 = (uint32_t)(sx0 + (int)ln * kLanePx32);  // first pixel of the group (multiple of 4)
                atomicOr(plane + (uint64_t)row * (a.mpitch >> 2) + (x0 >> 5), cb << (x0 & 31u));
            }
        }
        qn = 0;
    };

    constexpr int kAhead = 2;
#pragma unroll
    for (int s = 0; s < kAhead; ++s) fetch(pre[s], s);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const RowRegsU32 r = pre[s];
        fetch(pre[(s + kAhead) % 7], s + kAhead);
        push(s, r);
    }

    for (int base = 6;; base += 7) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int s = (6 + t) % 7;
            const int sc = (s + 4) % 7;
            const int i = base + t;
            if (i >= total) goto rows_done;
            {
                const RowRegsU32 r = pre[s];
                fetch(pre[(s + kAhead) % 7], i + kAhead);
                push(s, r);

                uint32_t WX[4], WM[4];
                {
                    const uint32_t L1 = from_left(colX[1]), L2 = from_left(colX[2]), L3 = from_left(colX[3]);
                    const uint32_t R0 = from_right(colX[0]), R1 = from_right(colX[1]), R2 = from_right(colX[2]);
                    WX[0] = (L1 + L2 + L3) + (colX[0] + colX[1] + colX[2]) + colX[3];
                    WX[1] = WX[0] - L1 + R0;
                    WX[2] = WX[1] - L2 + R1;
                    WX[3] = WX[2] - L3 + R2;
                }
                {
                    const uint32_t L1 = from_left(colM[1]), L2 = from_left(colM[2]), L3 = from_left(colM[3]);
                    const uint32_t R0 = from_right(colM[0]), R1 = from_right(colM[1]), R2 = from_right(colM[2]);
                    WM[0] = (L1 + L2 + L3) + (colM[0] + colM[1] + colM[2]) + colM[3];
                    WM[1] = WM[0] - L1 + R0;
                    WM[2] = WM[1] - L2 + R1;
                    WM[3] = WM[2] - L3 + R2;
                }
                const int yout = __builtin_amdgcn_readfirstlane(yb0 + (i - 6));
                __builtin_amdgcn_raw_buffer_store_b32(0u, r_sb, off_byte_st, (uint32_t)yout * a.bpitch, 0);

                // group screen: b_j <= m pmax - xmin for all four pixels when their windows hold the same
                // count; same float32 form and error bound as the per-pixel test
                const uint32_t fc = ringF[sc];
                const uint32_t xmin = min(min(WX[0], WX[1]), min(WX[2], WX[3]));
                const uint32_t mmin = min(min(WM[0], WM[1]), min(WM[2], WM[3]));
                const uint32_t mmax = max(max(WM[0], WM[1]), max(WM[2], WM[3]));
                const uint32_t pmax = max(max(ringX[sc][0], ringX[sc][1]), max(ringX[sc][2], ringX[sc][3]));
                const float xf = (float)xmin, mf = (float)mmin, pf = (float)pmax;
                const float sp = __builtin_fmaf(mf, pf, xf);
                const float bf = __builtin_fmaf(mf, pf, -xf);
                const float u = __builtin_fmaf(sp, 2.384185791015625e-07f, bf);
                const bool any_big = (((fc >> 4) & ~fc) & 0xFu) != 0u;
                const bool pass = (u * __builtin_fabsf(u) > kS * (xf * mf)) || mmin != mmax || any_big;
                const bool flag = owned && pass;
                const unsigned long long fm = __ballot(flag);
                if (fm) {  // wave-uniform
                    const int nf = __popcll(fm);
                    if (qn + nf > kQCap) drain();
                    const uint32_t QL1 = from_left(colQ[1]), QL2 = from_left(colQ[2]), QL3 = from_left(colQ[3]);
                    const uint32_t QR0 = from_right(colQ[0]), QR1 = from_right(colQ[1]), QR2 = from_right(colQ[2]);
                    if (flag) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        const int e = qn + (int)rank;
                        if (e >= kQCap) {
                            // queue full: the exact kernel takes the whole group
                            const uint32_t x0 = (uint32_t)cx;
                            atomicOr(plane + (uint64_t)yout * (a.mpitch >> 2) + (x0 >> 5), 0xFu << (x0 & 31u));
                        } else {
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                s_q[w][e] = WX[w];
                                s_q[4 + w][e] = WM[w];
                                s_q[8 + w][e] = ringX[sc][w];
                                s_q[16 + w][e] = colQ[w];
                            }
                            s_q[12][e] = fc;
                            s_q[13][e] = QL1; s_q[14][e] = QL2; s_q[15][e] = QL3;
                            s_q[20][e] = QR0; s_q[21][e] = QR1; s_q[22][e] = QR2;
                            s_q[23][e] = ((uint32_t)yout << 6) | (uint32_t)lane;
                        }
                    }
                    qn = min(qn + nf, kQCap);
                }
            }
        }
    }
rows_done:
    if (qn > 0) drain();
}

// ================================================================================================
// K2: exact predicate on candidates
// ================================================================================================

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

// extended algorithm's final test (kernels_extended.hpp)
template <typename PixelT>
__device__ bool ext_final_strong(const ThresholdArgs& a, const uint8_t* img, const uint8_t* eplane, int x, int y);

// MODE 0: candidates come from (and strong pixels go back to) a.bits, predicate exact_strong.
// MODE 1: extended algorithm -- candidates are the signal-region plane a.eplane (read-only: other
//         tiles read it for their 11x11 windows), predicate ext_final_strong, result in a.bits.
// MODE 2: extended algorithm's first pass after k_candidates_u16<true, true>: the pixels marked in
//         a.bits (positives the streaming kernel could not settle) take the exact dispersion test, the
//         failures are cleared in a.dplane; tiles without such pixels are left alone; no byte mask.
template <typename PixelT, int NT, int LISTCAP, int MODE = 0>
__device__ __forceinline__ void exact_tile(const ThresholdArgs& a) {
    // The stage is latency-bound (sparse gathers).  Measured dead ends: a smaller LDS footprint
    // (more tiles resident) and one-wave workgroups both made it slower.
    __shared__ uint32_t s_words[kTileRows * 320];  // tile bit-plane words (pitch_px <= 10240)
    __shared__ uint32_t s_plane[MODE == 2 ? kTileRows * 320 : 1];  // MODE 2: the dplane tile being edited
    __shared__ uint32_t s_list[LISTCAP];
    __shared__ uint32_t s_cnt, s_total, s_strong;

    const int tid = threadIdx.x;
    const int tile = blockIdx.x, frame = blockIdx.y;
    const int y0 = tile * kTileRows;
    const int rows = min(kTileRows, a.H - y0);
    const int dpr = a.mpitch >> 2;  // dwords per row
    const int ndw = rows * dpr;
    const uint8_t* img = (const uint8_t*)a.image + (uint64_t)frame * a.frame_stride;
    uint32_t* gwords = reinterpret_cast<uint32_t*>((MODE == 2 ? a.dplane : a.bits) + (uint64_t)frame * a.plane_frame_stride
                                                   + (uint64_t)y0 * a.mpitch);
    const uint8_t* eframe = MODE == 1 ? a.eplane + (uint64_t)frame * a.plane_frame_stride : nullptr;
    const uint32_t* gin = MODE == 1 ? reinterpret_cast<const uint32_t*>(eframe + (uint64_t)y0 * a.mpitch)
                        : MODE == 2 ? reinterpret_cast<const uint32_t*>(a.bits + (uint64_t)frame * a.plane_frame_stride
                                                                        + (uint64_t)y0 * a.mpitch)
                                    : gwords;
    uint8_t* sbytes = a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride;

    if (tid == 0) { s_cnt = 0; s_total = 0; s_strong = 0; }
    uint32_t mine = 0;
    for (int g = tid; g < ndw; g += NT) {
        const uint32_t w = gin[g];
        s_words[g] = w;
        mine += __popc(w);
    }
    __syncthreads();
    if (mine) atomicAdd(&s_total, mine);
    __syncthreads();
    const uint32_t total = s_total;  // block-uniform
    if (total == 0) {
        if (MODE != 2 && tid == 0) a.tile_counts[(uint64_t)frame * a.n_tiles + tile] = 0;
        if constexpr (MODE == 1)
            for (int g = tid; g < ndw; g += NT) gwords[g] = 0;
        return;
    }

    if constexpr (MODE == 2) {
        for (int g = tid; g < ndw; g += NT) s_plane[g] = gwords[g];
        __syncthreads();
    }

    auto append = [&](int g, uint32_t w, uint32_t at) {
        while (w) {
            const uint32_t bit = __ffs(w) - 1;
            w &= w - 1;
            s_list[at++] = ((uint32_t)g << 5) | bit;
        }
    };
    auto flush = [&]() {
        const uint32_t n = s_cnt;
        for (uint32_t e = tid; e < n; e += NT) {
            const uint32_t idx = s_list[e];
            const uint32_t g = idx >> 5, bit = idx & 31u;
            const int row = g / dpr;
            const int x = (int)((g - row * dpr) * 32u + bit);
            const int y = y0 + row;
            bool strong;
            if constexpr (MODE == 1) strong = ext_final_strong<PixelT>(a, img, eframe, x, y);
            else if constexpr (MODE == 2) strong = exact_strong<PixelT, true>(a, img, x, y);
            else strong = exact_strong<PixelT>(a, img, x, y);
            if (strong) {
                if constexpr (MODE != 2) sbytes[(uint64_t)y * a.bpitch + x] = 1;
            } else {
                atomicAnd(MODE == 2 ? &s_plane[g] : &s_words[g], ~(1u << bit));
            }
        }
    };

    if (total <= (uint32_t)LISTCAP) {
        // the usual case: every candidate of the tile in one dense pass
        if (mine) {
            uint32_t at = atomicAdd(&s_cnt, mine);
            for (int g = tid; g < ndw; g += NT) {
                const uint32_t w = s_words[g];
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
            const uint32_t w = (tid < LISTCAP / 32 && g < ndw) ? s_words[g] : 0u;
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
        const uint32_t w = MODE == 2 ? s_plane[g] : s_words[g];
        gwords[g] = w;
        cnt += __popc(w);
    }
    if (cnt) atomicAdd(&s_strong, cnt);
    __syncthreads();
    if (MODE != 2 && tid == 0) a.tile_counts[(uint64_t)frame * a.n_tiles + tile] = s_strong;
}

// NB: __launch_bounds__ must be a literal here -- with a template parameter hipcc 7.2 silently
// dropped it (default 1024-thread bound -> 178 VGPRs + scratch, kernel 2x slower).
template <typename PixelT>
__global__ __launch_bounds__(256) void k_exact(const ThresholdArgs a) { exact_tile<PixelT, 256, kExactListCap>(a); }
// one wave per tile: for the few candidates left after the dispersion screen
template <typename PixelT>
__global__ __launch_bounds__(64) void k_exact_w64(const ThresholdArgs a) { exact_tile<PixelT, 64, 256>(a); }
template __global__ void k_exact<uint16_t>(const ThresholdArgs);
template __global__ void k_exact<uint32_t>(const ThresholdArgs);
template __global__ void k_exact_w64<uint16_t>(const ThresholdArgs);
template __global__ void k_exact_w64<uint32_t>(const ThresholdArgs);
// extended first pass, exact stage (many more candidates per tile than the standard path)
template <typename PixelT>
__global__ __launch_bounds__(256) void k_exact_disp(const ThresholdArgs a) { exact_tile<PixelT, 256, kExactListCap, 2>(a); }
template __global__ void k_exact_disp<uint16_t>(const ThresholdArgs);
}  // namespace ffsamd
