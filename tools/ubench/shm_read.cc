// shm_read.cc -- how fast worker threads can pull 7.5 MB chunks out of a tmpfs directory: read() against mmap + memcpy,
// by number of threads.   g++ -O2 -std=c++17 -o shm_read shm_read.cc -lpthread && ./shm_read <dir with image_%06d_2 files> <n_files>
#ifdef WITH_HIP
#include <hip/hip_runtime.h>
#endif
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm/x";
    const int n_files = argc > 2 ? atoi(argv[2]) : 1000;
#ifdef WITH_HIP
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    const int n_modes = 8;   // 6: mmap + non-temporal copy beside the DMA; 7: read() through a cache-resident bounce buffer + non-temporal copy beside the DMA
#else
    const int n_modes = 3;
#endif
    for (int mode = 0; mode < n_modes; ++mode)
        for (int nt : {1, 4, 8, 16, 32}) {
            std::atomic<int> next{0};
            std::atomic<size_t> bytes{0};
#ifdef WITH_HIP
            std::atomic<bool> stop_dma{false};
            std::vector<char*> bufs(nt, nullptr);
            std::thread dma;
            if (mode >= 5) {
                dma = std::thread([&] {
                    (void)hipSetDevice(0);
                    void* d = nullptr;
                    (void)hipMalloc(&d, 40u << 20);
                    hipStream_t st;
                    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                    size_t moved = 0;
                    const auto t0 = std::chrono::steady_clock::now();
                    while (!stop_dma.load()) {
                        for (int t = 0; t < nt; ++t)
                            if (bufs[t]) { (void)hipMemcpyAsync(d, bufs[t], 30u << 20, hipMemcpyHostToDevice, st); moved += 30u << 20; }
                        (void)hipStreamSynchronize(st);
                    }
                    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    printf("   (DMA beside it: %.1f GB/s)\n", moved / s / 1e9);
                    (void)hipFree(d);
                });
            }
#endif
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t] {
                    const size_t cap = 40u << 20;
                    char* buf = (char*)mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    madvise(buf, cap, MADV_HUGEPAGE);
                    memset(buf, 0, cap);
#ifdef WITH_HIP
                    if (mode == 3 || mode >= 5) (void)hipHostRegister(buf, cap, hipHostRegisterDefault);
                    if (mode >= 5) bufs[t] = buf;
                    char* bounce = (char*)aligned_alloc(4096, 256u << 10);
                    auto nt_copy = [](char* d, const char* s_, size_t n) {   // d 32-byte aligned; n rounded up to 32
                        for (size_t o = 0; o < n; o += 32) _mm256_stream_si256((__m256i*)(d + o), _mm256_loadu_si256((const __m256i*)(s_ + o)));
                    };
                    char* hb = nullptr;
                    if (mode == 4) { (void)hipHostMalloc((void**)&hb, cap, 0); buf = hb; }
#endif
                    size_t got_all = 0, at = 0;
                    for (;;) {
                        const int i = next.fetch_add(1);
                        if (i >= n_files) break;
                        char name[64];
                        snprintf(name, sizeof name, "/image_%06d_2", i);
                        const int fd = open((dir + name).c_str(), O_RDONLY);
                        if (fd < 0) continue;
                        if (at + (12u << 20) > cap) at = 0;
                        at = (at + 63) & ~(size_t)63;
                        if (mode == 7) {
                            size_t got = 0;
                            for (;;) {
                                const ssize_t r = read(fd, bounce, 256u << 10);
                                if (r <= 0) break;
                                nt_copy(buf + at + got, bounce, ((size_t)r + 31) & ~(size_t)31);
                                got += (size_t)r;
                            }
                            _mm_sfence();
                            got_all += got; at += got;
                        } else if (mode == 6) {
                            struct stat sb; fstat(fd, &sb);
                            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
                            nt_copy(buf + at, (const char*)m, (size_t)sb.st_size & ~(size_t)31);
                            _mm_sfence();
                            munmap(m, (size_t)sb.st_size);
                            got_all += (size_t)sb.st_size; at += (size_t)sb.st_size;
                        } else if (mode == 0 || mode >= 3) {
                            size_t got = 0;
                            for (;;) { const ssize_t r = read(fd, buf + at + got, (12u << 20) - got); if (r <= 0) break; got += (size_t)r; }
                            got_all += got; at += got;
                        } else {
                            struct stat sb; fstat(fd, &sb);
                            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0);
                            memcpy(buf + at, m, (size_t)sb.st_size);
                            munmap(m, (size_t)sb.st_size);
                            got_all += (size_t)sb.st_size; at += (size_t)sb.st_size;
                        }
                        close(fd);
                    }
                    bytes += got_all;
#ifdef WITH_HIP
                    if (mode >= 5) return;   // (the DMA thread may still be reading from it: left mapped)
                    if (mode == 3) (void)hipHostUnregister(buf);
                    if (mode == 4) { (void)hipHostFree(hb); return; }
#endif
                    munmap(buf, cap);
                });
            for (auto& t : th) t.join();
#ifdef WITH_HIP
            if (mode >= 5) { stop_dma = true; dma.join(); }
#endif
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%-22s %2d threads: %.1f GB/s (%.0f files/s)\n", mode == 0 ? "read()" : mode == 1 ? "mmap + memcpy" : mode == 2 ? "mmap(POPULATE) + memcpy" : mode == 3 ? "read() -> registered" : mode == 4 ? "read() -> hipHostMalloc" : mode == 5 ? "read() -> registered + DMA" : mode == 6 ? "mmap + NT copy + DMA" : "read() bounce + NT copy + DMA", nt,
                   bytes.load() / s / 1e9, n_files / s);
        }
    return 0;
}
