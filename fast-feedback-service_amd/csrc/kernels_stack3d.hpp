// kernels_stack3d.hpp (included by ffs_stack3d.hip only) -- rotation sweeps: 3D connected components on the device.
//
// What the reference does (spotfinder/connected_components/connected_components.cc:270-470): after the last
// frame, one host thread copies every slice's 2D graph into one Boost graph, adds an edge for every linear
// index present in slices z and z + 1, labels the components, walks the slices in z order to fill
// Reflection3D objects and filters them.
//
// Here the per-frame strong-pixel lists never leave the GPU: ffs_stack3d_add_batch appends them to the
// stack's device buffers (k_stack_append), ffs_stack3d_finish puts the slices in frame order
// (k_stack_gather: also each entry's z, its parent and a fresh accumulator), then the same kernels as for
// single frames run over the whole stack as ONE segment -- k_union<true> (in-plane edges + the same pixel in
// the next slice), k_reduce_roots<true>, k_finalize_roots<true> (accumulators at the root's list index,
// records in (z, k) order of the roots = the order of Boost's labels) -- and k_stack_labels writes, for
// every strong pixel, its coordinates and the number of its component (the reference's signals_ view, used
// for the Kabsch-space variances).  Everything on the stack's own stream; no device-wide synchronisation.
#pragma once
#include "kernels_uf.hpp"

namespace ffsamd {

// grid (blocks, n_frames): frame f's list (src_k + f * src_stride ...) -> arrival buffers at table[f].dst
__global__ __launch_bounds__(256) void k_stack_append(const uint32_t* src_k, const uint32_t* src_i, uint64_t src_stride,
                                                      const StackSlice* table, uint32_t* dst_k, uint32_t* dst_i) {
    const StackSlice t = table[blockIdx.y];
    const uint32_t* sk = src_k + (uint64_t)blockIdx.y * src_stride;
    const uint32_t* si = src_i + (uint64_t)blockIdx.y * src_stride;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < t.n; e += gridDim.x * 256) {
        dst_k[t.dst + e] = sk[e];
        dst_i[t.dst + e] = si[e];
    }
}

// grid (blocks, n_slices): slice in arrival order -> its place in (z, k) order; z, parent and accumulator of every entry
__global__ __launch_bounds__(256) void k_stack_gather(const uint32_t* ak, const uint32_t* ai, const StackSlice* table,
                                                      uint32_t* k, uint32_t* inten, uint32_t* zs, uint32_t* parent, CompAcc* acc) {
    const StackSlice t = table[blockIdx.y];
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < t.n; e += gridDim.x * 256) {
        const uint32_t i = t.dst + e;
        k[i] = ak[t.src + e];
        inten[i] = ai[t.src + e];
        zs[i] = t.z;
        parent[i] = i;
        CompAcc a;
        a.sum_i = a.sum_xi = a.sum_yi = a.sum_zi = 0ull;
        a.peak = 0ull;
        a.x_min = 0xFFFFFFFFu; a.x_max = 0u;
        a.y_min = 0xFFFFFFFFu; a.y_max = 0u;
        a.z_min = 0x7FFFFFFF; a.z_max = (int32_t)0x80000000;
        a.num_pixels = 0u;
        a.root = i;
        acc[i] = a;
    }
}

// 3D reduction with the accumulators at the root: every entry on its own (one strong pixel), z from zs[]
__global__ __launch_bounds__(256) void k_reduce_roots3d(const SegArgs a) {
    __shared__ uint32_t s_nroots;
    const uint32_t n = min(a.seg_n[0], (uint32_t)a.seg_stride);
    const uint32_t* k = a.list_k;
    const uint32_t* inten = a.list_i;
    uint32_t* parent = a.parent;
    CompAcc* acc = a.acc;
    const int tid = threadIdx.x;
    for (uint32_t base = blockIdx.x * kRootChunk; base < n; base += gridDim.x * kRootChunk) {
        if (tid == 0) s_nroots = 0;
        __syncthreads();
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t i = base + (uint32_t)tid + 256u * q;
            if (i >= n) continue;
            const uint32_t ri = uf_find(parent, i);
            if (ri == i) ++mine;
            const uint32_t ki = k[i];
            const uint32_t y = ki / a.W, x = ki - y * a.W, z = a.zs[i];
            const unsigned long long I = inten[i];
            CompAcc* r = acc + ri;
            atomicMin(&r->x_min, x); atomicMax(&r->x_max, x);
            atomicMin(&r->y_min, y); atomicMax(&r->y_max, y);
            atomicMin(&r->z_min, (int32_t)z); atomicMax(&r->z_max, (int32_t)z);
            atomicAdd(&r->num_pixels, 1u);
            atomicAdd(&r->sum_i, I);
            atomicAdd(&r->sum_xi, (2ull * x + 1ull) * I);
            atomicAdd(&r->sum_yi, (2ull * y + 1ull) * I);
            atomicAdd(&r->sum_zi, (2ull * z + 1ull) * I);
            // highest intensity, ties -> smallest (z, y, x) = smallest list index
            // (connected_components.hpp:125-170, connected_components.cc:143-157)
            atomicMax(&r->peak, (I << 32) | (unsigned long long)(0xFFFFFFFFu - i));
        }
        if (mine) atomicAdd(&s_nroots, mine);
        __syncthreads();
        if (tid == 0) {
            a.chunk_roots[base / kRootChunk] = s_nroots;
            if (s_nroots) atomicAdd(&a.n_comp[0], s_nroots);
        }
        __syncthreads();
    }
}

// Records of the roots in list order (= label order), written to device memory; comp_id[root] = its number
__global__ __launch_bounds__(256) void k_finalize_roots3d(const SegArgs a) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_before;
    const int tid = threadIdx.x;
    const uint32_t n = min(a.seg_n[0], (uint32_t)a.seg_stride);
    const uint32_t chunks = (n + kRootChunk - 1) / kRootChunk;
    const uint32_t* k = a.list_k;
    const uint32_t* parent = a.parent;
    const CompAcc* acc = a.acc;
    ReflOut* recs = reinterpret_cast<ReflOut*>(a.recs);
    uint32_t* sm = a.summary;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        {
            uint32_t part = 0;
            for (uint32_t t = tid; t < c; t += 256) part += a.chunk_roots[t];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
            __syncthreads();
            if ((tid & 63) == 0) s_wave[tid >> 6] = part;
            __syncthreads();
            if (tid == 0) s_before = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            __syncthreads();
        }
        const uint32_t before = s_before;
        const uint32_t i0 = c * kRootChunk + 2u * (uint32_t)tid;
        const bool r0 = i0 < n && parent[i0] == i0, r1 = i0 + 1 < n && parent[i0 + 1] == i0 + 1;
        uint32_t total;
        const uint32_t rank = block_exclusive_scan<256>((r0 ? 1u : 0u) + (r1 ? 1u : 0u), s_wave, total);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!(q == 0 ? r0 : r1)) continue;
            const uint32_t i = i0 + q;
            const uint32_t cidx = before + rank + (q == 1 && r0 ? 1u : 0u);
            a.comp_id[i] = cidx;
            if (cidx >= a.max_comp) continue;
            const CompAcc r = acc[i];
            ReflOut o;
            o.x_min = r.x_min; o.x_max = r.x_max; o.y_min = r.y_min; o.y_max = r.y_max;
            o.z_min = r.z_min; o.z_max = r.z_max;
            o.num_pixels = (int32_t)r.num_pixels;
            o.sum_intensity = r.sum_i;
            // center_of_mass(): double sums of (c + 0.5) * I, quotient narrowed to float (connected_components.hpp:81-100)
            const double tot = (double)r.sum_i;
            o.com_x = (float)((double)r.sum_xi * 0.5 / tot);
            o.com_y = (float)((double)r.sum_yi * 0.5 / tot);
            o.com_z = (float)((double)r.sum_zi * 0.5 / tot);
            const uint32_t pi = min(0xFFFFFFFFu - (uint32_t)(r.peak & 0xFFFFFFFFull), n - 1);
            const uint32_t pk = k[pi];
            o.peak_y = pk / a.W;
            o.peak_x = pk - o.peak_y * a.W;
            o.peak_z = (int32_t)a.zs[pi];
            o.peak_intensity = (uint32_t)(r.peak >> 32);
            // peak_centroid_distance(): float arithmetic, one rounding per operation (connected_components.hpp:194-198)
            const float dx = ((float)o.peak_x + 0.5f) - o.com_x;
            const float dy = ((float)o.peak_y + 0.5f) - o.com_y;
            const float dz = ((float)o.peak_z + 0.5f) - o.com_z;
            const float s2 = (dx * dx + dy * dy) + dz * dz;
            o.peak_centroid_distance = (float)__builtin_sqrt((double)s2);
            uint32_t flags = 0;
            // filter_reflections(): size first, then separation (connected_components.cc:207-236)
            if (a.min_spot_size > 0 && r.num_pixels < a.min_spot_size) flags |= 1u;
            else if (a.max_sep > 0.0f && o.peak_centroid_distance > a.max_sep) flags |= 2u;
            o.flags = flags;
            recs[cidx] = o;
            if (flags == 0) atomicAdd(&sm[2], 1u);
            if (flags & 1u) atomicAdd(&sm[3], 1u);
            if (flags & 2u) atomicAdd(&sm[4], 1u);
        }
        __syncthreads();
    }
}

// Per strong pixel: x, y and the number of its component (label order, before filtering)
__global__ __launch_bounds__(256) void k_stack_labels(const SegArgs a, uint32_t* sig_x, uint32_t* sig_y, uint32_t* sig_comp) {
    const uint32_t n = min(a.seg_n[0], (uint32_t)a.seg_stride);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t ki = a.list_k[i];
        const uint32_t y = ki / a.W;
        sig_y[i] = y;
        sig_x[i] = ki - y * a.W;
        sig_comp[i] = a.comp_id[uf_find(a.parent, i)];
    }
}

}  // namespace ffsamd
