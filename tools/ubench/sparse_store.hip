// sparse_store.hip -- do scattered one-byte stores (the streaming kernel's plane bytes) cost the same in every allocation?
//   hipcc -O2 --offload-arch=gfx950 -o sparse_store sparse_store.hip && ./sparse_store
// N slabs of 2.33 GB are allocated one after the other (as the streams of 16 contexts are); in each, a region of 74 MB at
// the offset the bit plane has in a stream's slab takes 576 k byte stores at pseudo-random places from 15 000 waves, beside
// a streaming read of a 1.16 GB buffer by the same waves (so that the stores compete with loads as in the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(64, 4) void k_stores(const uint4* __restrict__ img, size_t n16, uint8_t* plane, size_t plane_bytes, uint32_t* sink,
                                                 int stores_per_wave, int with_stores, int flavour = 0) {
    const size_t wave = blockIdx.x, lane = threadIdx.x;
    const size_t per_wave = n16 / gridDim.x;
    const uint4* p = img + wave * per_wave;
    uint32_t acc = 0;
    uint32_t rng = (uint32_t)(wave * 2654435761u) ^ (uint32_t)lane * 40503u;
    const int every = (int)(per_wave / 64) / (stores_per_wave > 0 ? stores_per_wave : 1);
    int k = 0;
    uint32_t later[16]; int nl = 0;
    for (size_t i = lane; i < per_wave; i += 64, ++k) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (with_stores && every > 0 && (k % every) == 0 && lane < 3) {   // a few lanes store, as in a drain
            rng = rng * 1664525u + 1013904223u;
            const size_t at = (size_t)(rng % (uint32_t)plane_bytes);
            if (flavour == 4) { if (nl < 16) later[nl++] = (uint32_t)at; }
            else if (flavour == 0) plane[at] = (uint8_t)(acc | 1u);
            else if (flavour == 1) atomicOr(reinterpret_cast<uint32_t*>(plane) + (at >> 2), 1u << (at & 31u));
            else if (flavour == 2) __builtin_nontemporal_store((uint8_t)(acc | 1u), plane + at);
            else reinterpret_cast<uint32_t*>(plane)[at >> 2] = acc | 1u;
        }
    }
    for (int q = 0; q < nl; ++q) plane[later[q]] = (uint8_t)(acc | 1u);
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void k_stores_only(uint8_t* plane, size_t plane_bytes, int per_lane) {
    uint32_t rng = (uint32_t)(blockIdx.x * 2654435761u) ^ (uint32_t)threadIdx.x * 40503u;
    for (int k = 0; k < per_lane; ++k) {
        rng = rng * 1664525u + 1013904223u;
        plane[(size_t)(rng % (uint32_t)plane_bytes)] = (uint8_t)(rng | 1u);
    }
}
int main() {
    const size_t img_bytes = (size_t)32 * 4362 * 8320, plane_bytes = (size_t)32 * 4362 * 528, slab = (size_t)2330 << 20;
    uint4* img; uint32_t* sink;
    hipMalloc(&img, img_bytes); hipMemset(img, 1, img_bytes); hipMalloc(&sink, 256);
    const int N = 16;
    std::vector<uint8_t*> slabs(N);
    for (int i = 0; i < N; ++i) { if (hipMalloc(&slabs[i], slab) != hipSuccess) { printf("alloc %d failed\n", i); return 1; } hipMemset(slabs[i] + img_bytes, 0, plane_bytes); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round)
        for (int i = 0; i < N; ++i) {
            float ms[2];
            for (int ws = 0; ws < 2; ++ws) {
                hipEventRecord(e0);
                for (int r = 0; r < 10; ++r)
                    hipLaunchKernelGGL(k_stores, dim3(15064), dim3(64), 0, 0, img, img_bytes / 16, slabs[i] + img_bytes, plane_bytes, sink, 13, ws);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[ws], e0, e1);
            }
            if (round) printf("round %d slab %2d at %p: reads only %.1f us, with 576 k byte stores %.1f us (+%.1f)\n", round, i, (void*)slabs[i], ms[0] * 100, ms[1] * 100, (ms[1] - ms[0]) * 100);
        }
    // stores as they come against the same stores held back to the end of the wave (nothing left to wait for behind them)
    for (int i = 0; i < N; ++i) {
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0);
            for (int r = 0; r < 10; ++r)
                hipLaunchKernelGGL(k_stores, dim3(15064), dim3(64), 0, 0, img, img_bytes / 16, slabs[i] + img_bytes, plane_bytes, sink, 13, 1, v ? 4 : 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[v], e0, e1);
        }
        printf("slab %2d: stores as they come %.1f us, held back to the end of the wave %.1f us\n", i, ms[0] * 100, ms[1] * 100);
    }
    return 0;
}
