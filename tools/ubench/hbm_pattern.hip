// hbm_pattern.hip -- what the MI355X memory system delivers for the access patterns of the threshold
// kernel, without its arithmetic: 32 Eiger-16M frames (u16, pitch 8448 B) are read and a byte mask
// (pitch 4224 B) is written.
//   hipcc -O3 --offload-arch=gfx950 hbm_pattern.hip -o hbm_pattern && ./hbm_pattern
// Patterns:
//   linear      every wave reads consecutive 1 KiB pieces (grid-stride), writes consecutive 512 B pieces
//   march<D>    a wave owns a 1 KiB-wide column strip of one frame and walks down a band of rows with D rows
//               of loads in flight (the threshold kernel's structure), zero-filling 512 B of the byte mask per row
//   flags: W = with the byte-mask stores, N = non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int W = 4148, H = 4362, PITCH_PX = 4224, NF = 32;
constexpr uint32_t PITCH = PITCH_PX * 2, BPITCH = PITCH_PX;
constexpr uint64_t FSTRIDE = (uint64_t)PITCH * H, BFSTRIDE = (uint64_t)BPITCH * H;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

template <bool WR, bool NT>
__global__ __launch_bounds__(256) void k_linear(const uint8_t* img, uint8_t* bytes, uint32_t* sink, uint64_t n16) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(img)[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (WR) {
            uint2* p = reinterpret_cast<uint2*>(bytes) + i;
            if (NT) __builtin_nontemporal_store(make_uint2(0, 0).x, &p->x), __builtin_nontemporal_store(0u, &p->y);
            else *p = make_uint2(0, 0);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// strip marching: grid = (strips * bands, frames); 64 lanes x 16 B per row
template <int D, bool WR, bool NT, int PARTIAL>
__global__ __launch_bounds__(64) void k_march(const uint8_t* img, uint8_t* bytes, uint8_t* plane, uint32_t* sink, int n_strips, int band_rows) {
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int strip = q % n_strips;
    const int band = xcd + 8 * (q / n_strips);
    const int frame = blockIdx.y;
    const int y0 = band * band_rows, y1 = min(y0 + band_rows, H);
    if (y0 >= H) return;
    const rsrc_t r_img = make_rsrc(img + (uint64_t)frame * FSTRIDE, (uint32_t)FSTRIDE);
    const rsrc_t r_sb = make_rsrc(bytes + (uint64_t)frame * BFSTRIDE, (uint32_t)BFSTRIDE);
    const rsrc_t r_pl = make_rsrc(plane + (uint64_t)frame * (BFSTRIDE / 8), (uint32_t)(BFSTRIDE / 8));
    const int x0 = strip * 496 - 8 + lane * 8;
    const bool active = x0 >= 0 && x0 + 8 <= PITCH_PX;
    const uint32_t off_px = active ? (uint32_t)x0 * 2u : 0x80000000u;
    const uint32_t zc = (uint32_t)strip * 512u + (uint32_t)lane * 8u;
    const uint32_t off_b = zc < BPITCH ? zc : 0x80000000u;
    const uint32_t off_p = (active && lane >= 1 && lane <= 62) ? (uint32_t)x0 >> 3 : 0x80000000u;
    u32x4 pre[D];
#pragma unroll
    for (int d = 0; d < D; ++d) pre[d] = __builtin_amdgcn_raw_buffer_load_b128(r_img, off_px, (uint32_t)min(y0 + d, H - 1) * PITCH, 0);
    uint32_t acc = 0;
    for (int y = y0; y < y1; y += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const u32x4 v = pre[d];
            pre[d] = __builtin_amdgcn_raw_buffer_load_b128(r_img, off_px, (uint32_t)min(y + d + D, H - 1) * PITCH, 0);
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
            const int yy = __builtin_amdgcn_readfirstlane(y + d);
            if (PARTIAL == 2) {
                if (WR && yy < y1) {
                    const uint32_t zc2 = (uint32_t)strip * 512u + (uint32_t)lane * 16u;
                    const uint32_t ob = (lane < 32 && zc2 < BPITCH) ? zc2 : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, r_sb, ob, (uint32_t)yy * BPITCH, NT ? 2 : 0);
                }
            } else if (PARTIAL == 3) {
                if (WR && yy < y1 && (d & 1) == 0) {
                    const uint32_t zc2 = (uint32_t)strip * 512u + (uint32_t)(lane & 31) * 16u;
                    const uint32_t ob = zc2 < BPITCH ? zc2 + (uint32_t)(lane >> 5) * BPITCH : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, r_sb, ob, (uint32_t)yy * BPITCH, NT ? 2 : 0);
                }
            } else
            if (WR && yy < y1) {
                if (NT) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, r_sb, off_b, (uint32_t)yy * BPITCH, 2);
                else __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, r_sb, off_b, (uint32_t)yy * BPITCH, 0);
            }
            if (PARTIAL == 1 && yy < y1) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, r_pl, off_p, (uint32_t)yy * (BPITCH / 8), 0);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int WIDTH, bool NT>
__global__ __launch_bounds__(256) void k_fill(uint8_t* bytes, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n / WIDTH; i += stride) {
        if (WIDTH == 16) { uint4* p = reinterpret_cast<uint4*>(bytes) + i; if (NT) { __builtin_nontemporal_store(0u, &p->x); __builtin_nontemporal_store(0u, &p->y); __builtin_nontemporal_store(0u, &p->z); __builtin_nontemporal_store(0u, &p->w);} else *p = make_uint4(0, 0, 0, 0); }
        if (WIDTH == 8) { uint2* p = reinterpret_cast<uint2*>(bytes) + i; *p = make_uint2(0, 0); }
        if (WIDTH == 4) { reinterpret_cast<uint32_t*>(bytes)[i] = 0; }
    }
}

template <typename F>
static float time_it(F&& launch, int iters) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < iters; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    uint8_t *img, *bytes, *plane;
    uint32_t* sink;
    (void)hipMalloc(&img, FSTRIDE * NF + 4096);
    (void)hipMalloc(&bytes, BFSTRIDE * NF + 4096);
    (void)hipMalloc(&plane, BFSTRIDE / 8 * NF + 4096);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(img, 1, FSTRIDE * NF);
    (void)hipMemset(bytes, 0, BFSTRIDE * NF);
    const double rd = (double)FSTRIDE * NF, wr = (double)BFSTRIDE * NF;
    auto report = [&](const char* name, float ms, bool with_w) {
        const double bytes_moved = rd + (with_w ? wr : 0.0);
        printf("%-44s %8.1f us   %6.2f TB/s moved (%s)\n", name, ms * 1e3, bytes_moved / ms / 1e9, with_w ? "read + write" : "read only");
        fflush(stdout);
    };
    const uint64_t n16 = FSTRIDE * NF / 16;
    for (int blocks : {2048, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "linear read, %d blocks", blocks);
        report(nm, time_it([&] { hipLaunchKernelGGL((k_linear<false, false>), dim3(blocks), dim3(256), 0, 0, img, bytes, sink, n16); }, 10), false);
        snprintf(nm, sizeof nm, "linear read + write (8 B/lane), %d blocks", blocks);
        report(nm, time_it([&] { hipLaunchKernelGGL((k_linear<true, false>), dim3(blocks), dim3(256), 0, 0, img, bytes, sink, n16); }, 10), true);
    }
    {
        const uint64_t nb = BFSTRIDE * NF;
        auto rep = [&](const char* name, float ms) { printf("%-44s %8.1f us   %6.2f TB/s written\n", name, ms * 1e3, (double)nb / ms / 1e9); fflush(stdout); };
        rep("fill 16 B/lane", time_it([&] { hipLaunchKernelGGL((k_fill<16, false>), dim3(4096), dim3(256), 0, 0, bytes, nb); }, 10));
        rep("fill 16 B/lane nt", time_it([&] { hipLaunchKernelGGL((k_fill<16, true>), dim3(4096), dim3(256), 0, 0, bytes, nb); }, 10));
        rep("fill 8 B/lane", time_it([&] { hipLaunchKernelGGL((k_fill<8, false>), dim3(4096), dim3(256), 0, 0, bytes, nb); }, 10));
        rep("fill 4 B/lane", time_it([&] { hipLaunchKernelGGL((k_fill<4, false>), dim3(4096), dim3(256), 0, 0, bytes, nb); }, 10));
        rep("hipMemsetAsync", time_it([&] { (void)hipMemsetAsync(bytes, 0, nb, 0); }, 10));
    }
    for (int band_rows : {78}) {
        const int n_strips = 9, n_bands = (H + band_rows - 1) / band_rows, bands8 = (n_bands + 7) / 8 * 8;
        const dim3 grid(n_strips * bands8, NF);
        char nm[96];
#define RUN(D, WR, NT, P, label)                                                                                          \
        snprintf(nm, sizeof nm, "march band %d, %d rows ahead, %s", band_rows, D, label);                                     \
        report(nm, time_it([&] { hipLaunchKernelGGL((k_march<D, WR, NT, P>), grid, dim3(64), 0, 0, img, bytes, plane, sink, n_strips, band_rows); }, 10), WR);
        RUN(2, false, false, 0, "read only")
        RUN(4, false, false, 0, "read only")
        RUN(8, false, false, 0, "read only")
        RUN(2, true, false, 0, "read + zero-fill")
        RUN(4, true, false, 0, "read + zero-fill")
        RUN(8, true, false, 0, "read + zero-fill")
        RUN(4, true, true, 0, "read + zero-fill nt")
        RUN(4, true, false, 1, "read + zero-fill + byte-wise plane")
        RUN(4, true, true, 2, "read + zero-fill nt b128 x 32 lanes")
        RUN(4, true, true, 3, "read + zero-fill nt b128, two rows per store")
        RUN(4, true, false, 3, "read + zero-fill b128, two rows per store")
#undef RUN
    }
    return 0;
}
