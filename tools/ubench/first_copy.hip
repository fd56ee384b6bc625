// first_copy.hip -- what the FIRST host-to-device copy out of a freshly registered staging area costs, and whether a small
// warm-up copy pays it.   hipcc -O2 --offload-arch=gfx950 -o first_copy first_copy.hip && ./first_copy
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    const size_t n = (size_t)152 << 20, chunk = (size_t)7500000;
    void* d = nullptr;
    (void)hipMalloc(&d, n);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    { void* w = nullptr; (void)hipHostMalloc(&w, 1 << 20, 0); (void)hipMemcpyAsync(d, w, 1 << 20, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); (void)hipHostFree(w); }
    for (int variant = 0; variant < 4; ++variant) {
        uint8_t* h = (uint8_t*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(h, n, MADV_HUGEPAGE);
        auto t0 = std::chrono::steady_clock::now();
        if (variant != 3) std::memset(h, 1, n);
        const double touch = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        (void)hipHostRegister(h, n, hipHostRegisterDefault);
        const double reg = ms_since(t0);
        double warm_call = 0, warm_done = 0;
        if (variant == 1 || variant == 2) {   // warm-up: 4 KB, or one byte of every 2 MB page
            t0 = std::chrono::steady_clock::now();
            if (variant == 1) (void)hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, st);
            else (void)hipMemcpy2DAsync(d, 64, h, (size_t)2 << 20, 64, n >> 21, hipMemcpyHostToDevice, st);
            warm_call = ms_since(t0);
            (void)hipStreamSynchronize(st);
            warm_done = ms_since(t0);
        }
        double call[3], done[3];
        for (int r = 0; r < 3; ++r) {
            t0 = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync((uint8_t*)d + r * chunk, h + r * chunk, chunk, hipMemcpyHostToDevice, st);
            call[r] = ms_since(t0);
            (void)hipStreamSynchronize(st);
            done[r] = ms_since(t0);
        }
        t0 = std::chrono::steady_clock::now();
        (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        const double whole = ms_since(t0);
        std::printf("%s: touch %.1f ms, register %.1f ms, warm-up call %.2f done %.2f ms | 7.5 MB copies: call/done %.2f/%.2f  %.2f/%.2f  %.2f/%.2f ms | whole 152 MB %.2f ms\n",
                    variant == 0 ? "no warm-up      " : variant == 1 ? "4 KB warm-up    " : variant == 2 ? "1 line per 2 MB " : "untouched pages ", touch, reg, warm_call, warm_done,
                    call[0], done[0], call[1], done[1], call[2], done[2], whole);
        (void)hipHostUnregister(h);
        munmap(h, n);
    }
    return 0;
}
