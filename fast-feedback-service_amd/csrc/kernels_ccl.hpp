// kernels_ccl.hpp (included by ffs_submit.hip only) -- strong-pixel compaction and 2D connected components as
// grid-wide kernels on gfx950: the path of frames too tall for k_frame_chain (kernels_chain.hpp) and the A/B
// partner of that kernel (ffs_ctx_set_tuning "sparse_stage" = 1).
//
// Replaces the reference's host stage (spotfinder/connected_components/connected_components.cc:17-139,207-266):
//   k_emit_list_w     : strong bit plane -> per-frame list sorted by linear index, runs linked, byte mask 1s
//   k_union<false>    : kernels_uf.hpp (vertical edges + the row-wrap edge)
//   k_reduce_roots    : per-component bbox / sums / peak, integer accumulators at the root's list index
//   k_finalize_roots  : centre of mass, peak-centroid distance, filters, 40-byte wire records
#pragma once
#include "kernels_uf.hpp"

namespace ffsamd {



// The same compaction with ONE WAVE per (tile, frame).  The stage is a chain of dependent memory round
// trips (tile counts, plane words -> pixels), so what matters is how many tiles are in flight and how short
// the chain is: 32 waves per CU instead of 8 workgroups, both first loads issued together, and no
// cross-lane scan per 64 words -- the few non-zero words of a tile (a few dozen of ~1000) are first packed
// into an LDS list (ballot + mbcnt, no round trip), then one scan per 64 list entries places their pixels.
// It also does what k_link_runs did: an entry whose left neighbour in the same row is strong points at it
// (plain store; parents are always smaller indices), so that k_union is left with the vertical edges and
// the reference's row-wrap edge ((W-1, y) -- (0, y+1), no row-end check, connected_components.cc:62-70).
// Sets the 1s of the byte mask and, for k_stream_u16, clears every plane word it has consumed (that kernel
// needs an all-zero plane).
constexpr int kEmitListCap = 256;  // non-zero plane words staged per flush
constexpr int kEmitInFlight = 8;   // plane words a lane has in flight

// the first kEmitInFlight * 64 plane words of a tile (the caller issues these loads as early as it can)
__device__ __forceinline__ void emit_preload(const CclArgs& a, int frame, int tile, int lane, uint32_t (&wv)[kEmitInFlight]) {
    const int y0 = tile * kTileRows;
    const int ndw = min(kTileRows, a.H - y0) * (int)(a.mpitch >> 2);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(a.bits + (uint64_t)frame * a.plane_frame_stride + (uint64_t)y0 * a.mpitch);
#pragma unroll
    for (int q = 0; q < kEmitInFlight; ++q) {
        const int g = q * 64 + lane;
        wv[q] = g < ndw ? words[g] : 0u;
    }
}

// One wave compacts one tile: list entries from tile_base on, row offsets, run links, fresh accumulators, the 1s
// of the byte mask, the plane words cleared.  s_g / s_w: kEmitListCap words each, s_rows: kTileRows words (this wave's).
template <typename PixelT>
__device__ __forceinline__ void emit_tile_w(const CclArgs& a, int frame, int tile, uint32_t tile_base, uint32_t count,
                                            uint32_t (&wv)[kEmitInFlight], uint32_t* s_g, uint32_t* s_w, uint32_t* s_rows, int lane) {
    const int y0 = tile * kTileRows;
    const int rows = min(kTileRows, a.H - y0);
    const int dpr = a.mpitch >> 2;
    const int ndw = rows * dpr;
    uint32_t* words = reinterpret_cast<uint32_t*>(a.bits + (uint64_t)frame * a.plane_frame_stride + (uint64_t)y0 * a.mpitch);
    uint32_t* row_off = a.row_off + (uint64_t)frame * (a.H + 1);
    if (count == 0) {  // wave-uniform: empty rows all start where the tile starts
        if (lane < rows) row_off[y0 + lane] = min(tile_base, a.cap);
        return;
    }
    if (lane < kTileRows) s_rows[lane] = 0;
    const uint8_t* img = (const uint8_t*)a.image + (uint64_t)frame * a.frame_stride;
    uint32_t* lk = a.list_k + (uint64_t)frame * a.cap;
    uint32_t* li = a.list_i + (uint64_t)frame * a.cap;
    uint32_t* par = a.parent + (uint64_t)frame * a.cap;
    uint8_t* sbytes = a.strong_bytes + (uint64_t)frame * a.bytes_frame_stride;
    CompAcc2* acc2 = a.acc2 ? a.acc2 + (uint64_t)frame * a.cap : nullptr;

    uint32_t run = tile_base;              // list position of the next strong pixel (wave-uniform)
    int n_list = 0;                        // staged non-zero words (wave-uniform)
    uint32_t last_g = 0xFFFFFFFFu, last_w = 0;  // the last word of the previous flush (for the link across flushes)
    auto flush = [&]() {
        for (int base = 0; base < n_list; base += 64) {
            const int e = base + lane;
            const bool valid = e < n_list;
            const uint32_t g = valid ? s_g[e] : 0u;
            uint32_t w = valid ? s_w[e] : 0u;
            const uint32_t pg = e > 0 ? s_g[valid ? e - 1 : 0] : last_g, pw = e > 0 ? s_w[valid ? e - 1 : 0] : last_w;
            const uint32_t pc = (uint32_t)__popc(w);
            uint32_t inc = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            uint32_t at = run + inc - pc;
            run += __shfl(inc, 63, 64);
            if (valid) {
                const int row = (int)g / dpr;
                const int col = (int)g - row * dpr;
                atomicAdd(&s_rows[row], pc);
                const int xb = col * 32;
                const int y = y0 + row;
                // the pixel left of this word's bit 0: bit 31 of the word before it, same row
                const bool left = col != 0 && pg + 1 == g && (pw >> 31) != 0u;
                int prev = -2;  // bit index of this word's previous strong pixel
                uint32_t run_first = at;  // list index of the first pixel of the current run inside this word
                while (w) {
                    const int bit = __ffs((int)w) - 1;
                    w &= w - 1;
                    const int x = xb + bit;
                    const bool linked = bit == 0 ? left : prev == bit - 1;
                    if (!linked || bit == 0) run_first = at;
                    if (at < a.cap) {
                        lk[at] = (uint32_t)y * (uint32_t)a.W + (uint32_t)x;
                        li[at] = *reinterpret_cast<const PixelT*>(img + (uint64_t)y * a.pitch + (uint64_t)x * sizeof(PixelT));
                        // a run's pixels point at its first pixel in this word (a find is one step, not a walk
                        // along the run); a run that continues from the previous word hooks on to that word's last pixel
                        par[at] = !linked ? at : (bit == 0 ? at - 1 : run_first);
                        if (acc2 && !linked) {  // a run start may end up a root: fresh accumulator
                            CompAcc2 z;
                            z.sum_i = z.sum_xi = z.sum_yi = z.peak = 0ull;
                            z.x_min = 0xFFFFFFFFu; z.x_max = 0u; z.y_min = 0xFFFFFFFFu; z.y_max = 0u;
                            z.num_pixels = 0u; z.pad = 0u;
                            acc2[at] = z;
                        }
                    }
                    if (a.dense_bytes) sbytes[(uint64_t)y * a.bpitch + (uint32_t)x] = 1;  // the reference kernel's result_strong byte
                    ++at;
                    prev = bit;
                }
            }
        }
        if (n_list > 0) {
            last_g = s_g[n_list - 1];
            last_w = s_w[n_list - 1];
        }
        n_list = 0;
    };

    for (int c0 = 0; c0 * 64 < ndw; c0 += kEmitInFlight) {
        if (c0 > 0) {
#pragma unroll
            for (int q = 0; q < kEmitInFlight; ++q) {
                const int g = (c0 + q) * 64 + lane;
                wv[q] = g < ndw ? words[g] : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < kEmitInFlight; ++q) {
            if ((c0 + q) * 64 >= ndw) break;  // wave-uniform
            const uint32_t w = wv[q];
            const unsigned long long nz = __builtin_amdgcn_ballot_w64(w != 0u);
            if (nz == 0ull) continue;  // wave-uniform
            if (n_list + 64 > kEmitListCap) flush();
            if (w) {
                const int g = (c0 + q) * 64 + lane;
                if (a.clear_bits) words[g] = 0;
                const int e = n_list + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));
                s_g[e] = (uint32_t)g;
                s_w[e] = w;
            }
            n_list += __popcll(nz);
        }
    }
    flush();
    // list offset of the first strong pixel of every image row of the tile
    if (lane < rows) {
        uint32_t before = 0;
        for (int r = 0; r < lane; ++r) before += s_rows[r];
        row_off[y0 + lane] = min(tile_base + before, a.cap);
    }
}

template <typename PixelT>
__global__ __launch_bounds__(64) void k_emit_list_w(const CclArgs a) {
    __shared__ uint32_t s_g[kEmitListCap], s_w[kEmitListCap];
    __shared__ uint32_t s_rows[kTileRows];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x, frame = blockIdx.y;
    const uint32_t* counts = a.tile_counts + (uint64_t)frame * a.n_tiles;
    // first round trip: this tile's count, the counts of the tiles before it, and the first plane words
    const uint32_t count = counts[tile];
    uint32_t part = 0;
    for (int t = lane; t < tile; t += 64) part += counts[t];
    uint32_t wv[kEmitInFlight];
    emit_preload(a, frame, tile, lane, wv);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    const uint32_t tile_base = part;
    if (tile == a.n_tiles - 1 && lane == 0) {  // the last tile knows the frame's total
        const uint32_t total = tile_base + count;
        a.num_strong[frame] = total;
        (a.row_off + (uint64_t)frame * (a.H + 1))[a.H] = min(total, a.cap);
        if (total > a.cap) atomicOr(a.overflow, 1u);
    }
    if (tile == 0 && a.acc2) {  // counters the root-indexed reduction adds into
        if (lane == 0) a.n_comp[frame] = 0;
        if (lane < 8) a.summary[(uint64_t)frame * 8 + lane] = 0;
    }
    emit_tile_w<PixelT>(a, frame, tile, tile_base, count, wv, s_g, s_w, s_rows, lane);
}
template __global__ void k_emit_list_w<uint16_t>(const CclArgs);
template __global__ void k_emit_list_w<uint32_t>(const CclArgs);

// ---- 2D: reduction with the accumulators at the root (no numbering pass) ---------------------------------
// k_union leaves every component as a tree whose root is its smallest list index.  k_reduce_roots adds
// every horizontal run into the accumulator AT that index (LDS first when the root lies in the same chunk
// of kRootChunk entries) and counts the roots of each chunk; k_finalize_roots turns the roots into records
// in list order = label order (connected_components.cc:91,242: Boost numbers components by their first
// vertex), each chunk adding up the root counts of the chunks before it.  Replaces k_count_roots +
// k_label_parts + k_reduce + k_finalize for single frames.

struct LdsAcc2 {
    unsigned long long sum_i, sum_xi, sum_yi, peak;
    uint32_t x_min, x_max, y_min, y_max;
    uint32_t num_pixels, pad;
};

__global__ __launch_bounds__(256) void k_reduce_roots(const SegArgs a) {
    __shared__ LdsAcc2 s_acc[kRootChunk];
    __shared__ uint32_t s_nroots;
    const int seg = blockIdx.y;
    const uint32_t n = min(a.seg_n[seg], (uint32_t)a.seg_stride);
    const uint32_t* k = a.list_k + (uint64_t)seg * a.seg_stride;
    const uint32_t* inten = a.list_i + (uint64_t)seg * a.seg_stride;
    uint32_t* parent = a.parent + (uint64_t)seg * a.seg_stride;
    CompAcc2* acc = a.acc2 + (uint64_t)seg * a.seg_stride;
    const int tid = threadIdx.x;
    for (uint32_t base = blockIdx.x * kRootChunk; base < n; base += gridDim.x * kRootChunk) {
        if (tid == 0) s_nroots = 0;
        bool root[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t e = (uint32_t)tid + 256u * q, i = base + e;
            root[q] = i < n && parent[i] == i;
            if (root[q]) {
                LdsAcc2 z;
                z.sum_i = z.sum_xi = z.sum_yi = z.peak = 0ull;
                z.x_min = 0xFFFFFFFFu; z.x_max = 0u; z.y_min = 0xFFFFFFFFu; z.y_max = 0u;
                z.num_pixels = 0; z.pad = 0;
                s_acc[e] = z;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t e = (uint32_t)tid + 256u * q, i0 = base + e;
            // consecutive-k entries form a horizontal run = one component; the thread of a run's first entry
            // (runs are also cut every 32 entries and at the chunk start) sums the whole segment in registers
            const bool owner = i0 < n && ((e & 31u) == 0u || k[i0 - 1] + 1 != k[i0]);
            if (!owner) continue;
            uint32_t j = i0 + 1;
            while (j < n && (j - base) < (uint32_t)kRootChunk && ((j - base) & 31u) != 0u && k[j] == k[j - 1] + 1) ++j;
            const uint32_t ri = uf_find(parent, i0);
            uint32_t x_min = 0xFFFFFFFFu, x_max = 0u, y_min = 0xFFFFFFFFu, y_max = 0u, npx = 0u;
            unsigned long long s_i = 0, s_xi = 0, s_yi = 0, pk = 0;
            for (uint32_t i = i0; i < j; ++i) {
                const uint32_t ki = k[i];
                const uint32_t y = ki / a.W, x = ki - y * a.W;
                const unsigned long long I = inten[i];
                x_min = min(x_min, x); x_max = max(x_max, x);
                y_min = min(y_min, y); y_max = max(y_max, y);
                ++npx;
                s_i += I;
                s_xi += (2ull * x + 1ull) * I;
                s_yi += (2ull * y + 1ull) * I;
                // highest intensity, ties -> smallest (y, x) = smallest list index
                // (connected_components.hpp:125-170, connected_components.cc:143-157)
                pk = max(pk, (I << 32) | (unsigned long long)(0xFFFFFFFFu - i));
            }
            if (ri >= base) {  // root inside this chunk
                LdsAcc2* r = &s_acc[ri - base];
                atomicMin(&r->x_min, x_min); atomicMax(&r->x_max, x_max);
                atomicMin(&r->y_min, y_min); atomicMax(&r->y_max, y_max);
                atomicAdd(&r->num_pixels, npx);
                atomicAdd(&r->sum_i, s_i);
                atomicAdd(&r->sum_xi, s_xi);
                atomicAdd(&r->sum_yi, s_yi);
                atomicMax(&r->peak, pk);
            } else {
                CompAcc2* r = acc + ri;
                atomicMin(&r->x_min, x_min); atomicMax(&r->x_max, x_max);
                atomicMin(&r->y_min, y_min); atomicMax(&r->y_max, y_max);
                atomicAdd(&r->num_pixels, npx);
                atomicAdd(&r->sum_i, s_i);
                atomicAdd(&r->sum_xi, s_xi);
                atomicAdd(&r->sum_yi, s_yi);
                atomicMax(&r->peak, pk);
            }
        }
        __syncthreads();
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!root[q]) continue;
            ++mine;
            const uint32_t e = (uint32_t)tid + 256u * q;
            const LdsAcc2 v = s_acc[e];
            CompAcc2* r = acc + base + e;   // entries of later chunks add to it directly, hence atomics
            atomicMin(&r->x_min, v.x_min); atomicMax(&r->x_max, v.x_max);
            atomicMin(&r->y_min, v.y_min); atomicMax(&r->y_max, v.y_max);
            atomicAdd(&r->num_pixels, v.num_pixels);
            atomicAdd(&r->sum_i, v.sum_i);
            atomicAdd(&r->sum_xi, v.sum_xi);
            atomicAdd(&r->sum_yi, v.sum_yi);
            atomicMax(&r->peak, v.peak);
        }
        if (mine) atomicAdd(&s_nroots, mine);
        __syncthreads();
        if (tid == 0) {
            a.chunk_roots[(uint64_t)seg * a.chunks_max + base / kRootChunk] = s_nroots;
            if (s_nroots) atomicAdd(&a.n_comp[seg], s_nroots);
        }
        __syncthreads();
    }
}

// One workgroup per chunk (grid-strided): records of the chunk's roots, in list order, staged in LDS and
// written out as consecutive dwords (the record buffer usually is host memory behind PCIe: 40-byte wire
// records, whole lines).
__global__ __launch_bounds__(256) void k_finalize_roots(const SegArgs a) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_before, s_base;
    __shared__ uint32_t s_out[kRootChunk * (sizeof(WireRec2) / 4)];
    const int seg = blockIdx.y, tid = threadIdx.x;
    const uint32_t n = min(a.seg_n[seg], (uint32_t)a.seg_stride);
    const uint32_t chunks = (n + kRootChunk - 1) / kRootChunk;
    const uint32_t* k = a.list_k + (uint64_t)seg * a.seg_stride;
    const uint32_t* parent = a.parent + (uint64_t)seg * a.seg_stride;
    const CompAcc2* acc = a.acc2 + (uint64_t)seg * a.seg_stride;
    const uint32_t* croots = a.chunk_roots + (uint64_t)seg * a.chunks_max;
    uint32_t* sm = a.summary + (uint64_t)seg * 8;
    if (tid == 0) {
        uint32_t b = 0;
        for (int q = 0; q < seg; ++q) b += min(a.n_comp[q], a.max_comp);
        s_base = b;
        if (blockIdx.x == 0 && a.n_comp[seg] > a.max_comp) atomicOr(a.overflow, 2u);
    }
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        {   // roots in the chunks before this one
            uint32_t part = 0;
            for (uint32_t t = tid; t < c; t += 256) part += croots[t];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
            __syncthreads();
            if ((tid & 63) == 0) s_wave[tid >> 6] = part;
            __syncthreads();
            if (tid == 0) s_before = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            __syncthreads();
        }
        const uint32_t before = s_before;
        const uint32_t nroots = croots[c];
        // each thread owns two consecutive entries, so one block scan numbers the roots in list order
        const uint32_t i0 = c * kRootChunk + 2u * (uint32_t)tid;
        const bool r0 = i0 < n && parent[i0] == i0, r1 = i0 + 1 < n && parent[i0 + 1] == i0 + 1;
        uint32_t total;
        const uint32_t rank = block_exclusive_scan<256>((r0 ? 1u : 0u) + (r1 ? 1u : 0u), s_wave, total);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!(q == 0 ? r0 : r1)) continue;
            const uint32_t i = i0 + q;
            const uint32_t slot = rank + (q == 1 && r0 ? 1u : 0u);  // record number inside the chunk
            if (before + slot >= a.max_comp) continue;
            const CompAcc2 r = acc[i];
            // center_of_mass(): double sums of (c + 0.5) * I, quotient narrowed to float
            // (connected_components.hpp:81-100).  sum (2c+1) I is an exact integer; * 0.5 is exact.
            const double tot = (double)r.sum_i;
            const float com_x = (float)((double)r.sum_xi * 0.5 / tot), com_y = (float)((double)r.sum_yi * 0.5 / tot);
            const float com_z = (float)(0.5 * tot / tot);  // z = 0 for 2D (:247)
            const uint32_t pi = min(0xFFFFFFFFu - (uint32_t)(r.peak & 0xFFFFFFFFull), n - 1);  // (never clamps: every root owns a pixel)
            const uint32_t pk = k[pi];
            const uint32_t peak_y = pk / a.W, peak_x = pk - peak_y * a.W;
            // peak_centroid_distance(): float arithmetic, connected_components.hpp:194-198; one rounding per
            // operation (no contraction in this library), float sqrt through the correctly rounded double sqrt
            const float dx = ((float)peak_x + 0.5f) - com_x;
            const float dy = ((float)peak_y + 0.5f) - com_y;
            const float dz = ((float)0 + 0.5f) - com_z;
            const float s2 = (dx * dx + dy * dy) + dz * dz;
            const float pcd = (float)__builtin_sqrt((double)s2);
            uint32_t flags = 0;
            // filter_reflections(): size first, then separation (connected_components.cc:207-236)
            if (a.min_spot_size > 0 && r.num_pixels < a.min_spot_size) flags |= 1u;
            else if (a.max_sep > 0.0f && pcd > a.max_sep) flags |= 2u;
            WireRec2 o;
            o.x_min = (uint16_t)r.x_min; o.x_max = (uint16_t)r.x_max; o.y_min = (uint16_t)r.y_min; o.y_max = (uint16_t)r.y_max;
            o.npx_flags = r.num_pixels | (flags << 30);
            o.com_x = com_x; o.com_y = com_y;
            o.peak_x = (uint16_t)peak_x; o.peak_y = (uint16_t)peak_y;
            o.peak_intensity = (uint32_t)(r.peak >> 32);
            o.peak_centroid_distance = pcd;
            o.sum_intensity = r.sum_i;
            *reinterpret_cast<WireRec2*>(&s_out[slot * (sizeof(WireRec2) / 4)]) = o;
            // generate_boxes() filter (connected_components.cc:122-138)
            if (a.min_spot_size == 0 || r.num_pixels >= a.min_spot_size) {
                atomicAdd(&sm[0], 1u);
                atomicAdd(&sm[1], r.num_pixels);
            }
            if (flags == 0) atomicAdd(&sm[2], 1u);
            if (flags & 1u) atomicAdd(&sm[3], 1u);
            if (flags & 2u) atomicAdd(&sm[4], 1u);
        }
        __syncthreads();
        {
            const uint32_t first = min(before, a.max_comp), last = min(before + nroots, a.max_comp);
            uint32_t* dst = reinterpret_cast<uint32_t*>(reinterpret_cast<WireRec2*>(a.recs) + s_base + first);
            const uint32_t ndw = (last - first) * (uint32_t)(sizeof(WireRec2) / 4);
            for (uint32_t w = tid; w < ndw; w += 256) dst[w] = s_out[w];
        }
        __syncthreads();
    }
}

}  // namespace ffsamd
