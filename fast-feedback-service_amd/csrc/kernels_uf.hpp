// kernels_uf.hpp -- union-find on sorted strong-pixel lists and the block scan, shared by the 2D kernels
// (kernels_ccl.hpp, kernels_chain.hpp) and the 3D stack (kernels_stack3d.hpp).  Only templates and
// __device__ functions: this header may be included by several translation units.
//
// The reference builds a Boost adjacency_list with edges to k+1 and k+width (and the same pixel of the next
// slice in 3D) and runs boost::connected_components (DFS => components numbered by their minimum vertex):
// spotfinder/connected_components/connected_components.cc:47-79,91,352-370.  Here: lock-free union-find,
// union by minimum index (atomicMin hooks), neighbours found by binary search in the sorted list.
#pragma once
#include "ffs_device.h"

namespace ffsamd {

// ---- block-wide exclusive prefix sum (256 or 1024 threads) ---------------------------------------
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t c = s_wave[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// ---- compaction ------------------------------------------------------------------------------------

// One block per (tile, frame): bits -> (k, intensity) in raster order; parent[i] = i.
// The tile's offset in the frame's list is the sum of the counts of the tiles before it: every block
// adds them up itself (at most a few hundred words from L2) instead of waiting for a scan kernel.

// ---- union-find --------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t ld_parent(const uint32_t* p) {
    // agent-scope load: bypasses this CU's L1, which other CUs' atomics never refresh
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t v) {
    uint32_t p = ld_parent(parent + v);
    while (p != v) {
        v = p;
        p = ld_parent(parent + v);
    }
    return v;
}

// Union by minimum index: the root of every tree is its smallest member, so
// label order = order of the minimum vertex = Boost's DFS discovery order.
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        const uint32_t old = atomicMin(parent + b, a);  // hook the larger root under the smaller
        if (old == b) return;
        b = old;  // somebody else re-parented b meanwhile: retry from there
    }
}

// 2D pre-pass: entries whose left neighbour in the list is k - 1 belong to the same horizontal run (the
// reference's k + 1 edge, row wrap included).  Hook each of them to an earlier member of its run with
// a plain store -- no atomics, no contention -- so that k_union only has the vertical edges left, and
// of those only one per pair of overlapping runs.  (The backward walk is capped: pointing at any

template <bool IS3D>
__global__ __launch_bounds__(256) void k_union(const SegArgs a) {
    const int seg = blockIdx.y;
    const uint32_t n = min(a.seg_n[seg], (uint32_t)a.seg_stride);
    const uint32_t* k = a.list_k + (uint64_t)seg * a.seg_stride;
    uint32_t* parent = a.parent + (uint64_t)seg * a.seg_stride;
    if (!IS3D && a.zero_counts) {
        for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < a.zero_per_seg; t += gridDim.x * 256)
            a.zero_counts[(uint64_t)seg * a.zero_per_seg + t] = 0;
        if (seg == 0 && blockIdx.x == 0 && threadIdx.x == 0 && a.zero_word) *a.zero_word = 0;
    }
    uint32_t z = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint32_t s_end = n, nb = 0, ne = 0;
        if (IS3D) {
            // slice of entry i (slices ascending; advance monotonically within this thread)
            if (a.zs) z = a.zs[i];
            while (a.slice_begin[z + 1] <= i) ++z;
            s_end = a.slice_begin[z + 1];
            if ((int)z + 1 < a.n_slices) { nb = a.slice_begin[z + 1]; ne = a.slice_begin[z + 2]; }
        }
        const uint32_t ki = k[i];
        // right neighbour: k + 1, with NO row-end check (connected_components.cc:62-70).  2D: the compaction linked
        // the runs inside each image row; the edge from a row's last pixel to the next row's first is left to do
        if (IS3D) {
            if (i + 1 < s_end && k[i + 1] == ki + 1) uf_union(parent, i, i + 1);
        } else if (i > 0 && k[i - 1] + 1 == ki && ki % a.W == 0) {
            uf_union(parent, i - 1, i);
        }
        // neighbour below: k + width (:63, :73-78); it lives in the next image row, whose
        // list range is known from the compaction (row_off), so the search is a few steps
        {
            uint32_t lo = i + 1, hi = min(s_end, i + 1 + a.W);
            if (!IS3D && a.row_off) {
                const uint32_t* ro = a.row_off + (uint64_t)seg * (a.H + 1);
                const uint32_t y = ki / a.W;
                if (y + 1 < a.H) { lo = max(lo, ro[y + 1]); hi = min(hi, ro[y + 2]); } else hi = lo;
            }
            const uint32_t key = ki + a.W;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (k[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < s_end && k[lo] == key) {
                // with the runs linked, one edge per pair of overlapping runs is enough: the leftmost
                // overlapping pair has a run start on one side (if neither pixel starts its run, the
                // pair one column to the left is adjacent too)
                const bool needed = IS3D || i == 0 || k[i - 1] + 1 != ki || lo == 0 || k[lo - 1] + 1 != key;
                if (needed) uf_union(parent, i, lo);
            }
        }
        if (IS3D && nb < ne) {  // same pixel in the next slice (:352-370)
            uint32_t lo = nb, hi = ne;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (k[mid] < ki) lo = mid + 1; else hi = mid;
            }
            if (lo < ne && k[lo] == ki) uf_union(parent, i, lo);
        }
    }
}
template __global__ void k_union<false>(const SegArgs);
template __global__ void k_union<true>(const SegArgs);

constexpr int kRootChunk = 512;   // list entries per chunk of the root-indexed reductions

}  // namespace ffsamd
