// pin_cost.hip -- ways to get pinned (DMA-able) host memory and what each costs per GB on this box.
//   hipcc -O2 --offload-arch=gfx950 -o pin_cost pin_cost.hip && ./pin_cost
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static double h2d_gbps(void* h, void* d, size_t n, hipStream_t st) {
    double best = 1e9;
    for (int r = 0; r < 3; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        best = std::min(best, ms_since(t0));
    }
    return n / best / 1e6;
}
int main() {
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    const size_t n = (size_t)256 << 20;
    void* d = nullptr;
    (void)hipMalloc(&d, n);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    { void* w = nullptr; (void)hipHostMalloc(&w, 1 << 20, 0); (void)hipMemcpy(d, w, 1 << 20, hipMemcpyHostToDevice); (void)hipHostFree(w); }
    struct { const char* name; unsigned flags; } variants[] = {
        {"hipHostMalloc default", hipHostMallocDefault}, {"hipHostMalloc NonCoherent", hipHostMallocNonCoherent},
        {"hipHostMalloc Portable|Mapped", hipHostMallocPortable | hipHostMallocMapped}, {"hipHostMalloc WriteCombined", hipHostMallocWriteCombined},
        {"hipHostMalloc NumaUser", hipHostMallocNumaUser}};
    for (auto& v : variants) {
        void* h = nullptr;
        auto t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipHostMalloc(&h, n, v.flags);
        const double a = ms_since(t0);
        if (e != hipSuccess) { std::printf("%-32s failed: %s\n", v.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        t0 = std::chrono::steady_clock::now();
        std::memset(h, 1, n);
        const double touch = ms_since(t0);
        const double bw = h2d_gbps(h, d, n, st);
        t0 = std::chrono::steady_clock::now();
        (void)hipHostFree(h);
        std::printf("%-32s 256 MB: alloc %.1f ms, first touch %.1f ms, H2D %.1f GB/s, free %.1f ms\n", v.name, a, touch, bw, ms_since(t0));
    }
    for (int huge = 0; huge < 2; ++huge) {
        auto t0 = std::chrono::steady_clock::now();
        void* h = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (huge) madvise(h, n, MADV_HUGEPAGE);
        std::memset(h, 1, n);
        const double touch = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault);
        const double reg = ms_since(t0);
        if (e != hipSuccess) { std::printf("mmap%s + hipHostRegister failed: %s\n", huge ? " + MADV_HUGEPAGE" : "", hipGetErrorString(e)); continue; }
        const double bw = h2d_gbps(h, d, n, st);
        t0 = std::chrono::steady_clock::now();
        (void)hipHostUnregister(h);
        const double unreg = ms_since(t0);
        munmap(h, n);
        std::printf("mmap%-15s + touch %.1f ms, hipHostRegister %.1f ms, H2D %.1f GB/s, unregister %.1f ms\n", huge ? " + MADV_HUGEPAGE" : "", touch, reg, bw, unreg);
    }
    {   // unpinned source: what the runtime's own staging gives
        void* h = malloc(n);
        std::memset(h, 1, n);
        std::printf("pageable malloc, H2D %.1f GB/s\n", h2d_gbps(h, d, n, st));
        free(h);
    }
    return 0;
}
