// cold_read.cc -- the first pass over freshly written tmpfs files against later passes, by access method and threads.
//   g++ -O2 -mavx2 -std=c++17 -o cold_read cold_read.cc -lpthread && ./cold_read <dir> <n_files> <file_bytes>
// Every trial writes its own set of files (write(), one writer thread), then reads it `passes` times with `nt` threads.
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
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm/cold";
    const int n_files = argc > 2 ? atoi(argv[2]) : 400;
    const size_t fbytes = argc > 3 ? (size_t)atol(argv[3]) : (size_t)7500000;
    std::vector<char> payload(fbytes);
    for (size_t i = 0; i < fbytes; ++i) payload[i] = (char)(i * 2654435761u >> 24);
    int trial = 0;
    for (int method : {0, 4, 1, 5, 3})       // 0 read(), 1 mmap + memcpy, 2 mmap(POPULATE) + memcpy, 3 read() in 256 KB pieces through a bounce buffer + NT copy,
                                             // 4 read() after posix_fadvise(NOREUSE), 5 mmap + NT copy
        for (int nt : {8, 16}) {
            const std::string sub = dir + "/t" + std::to_string(trial++);
            mkdir(sub.c_str(), 0700);
            const double w0 = now();
            for (int i = 0; i < n_files; ++i) {
                const int fd = open((sub + "/f" + std::to_string(i)).c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0600);
                size_t put = 0;
                while (put < fbytes) { const ssize_t r = write(fd, payload.data() + put, fbytes - put); if (r <= 0) break; put += (size_t)r; }
                close(fd);
            }
            const double w1 = now();
            printf("method %d threads %2d: written at %.1f GB/s;", method, nt, n_files * (double)fbytes / (w1 - w0) / 1e9);
            for (int pass = 0; pass < 3; ++pass) {
                std::atomic<int> next{0};
                const double t0 = now();
                std::vector<std::thread> th;
                for (int t = 0; t < nt; ++t)
                    th.emplace_back([&] {
                        const size_t cap = 40u << 20;
                        char* buf = (char*)mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                        madvise(buf, cap, MADV_HUGEPAGE);
                        memset(buf, 0, cap);
                        char* bounce = (char*)aligned_alloc(4096, 256u << 10);
                        size_t at = 0;
                        for (;;) {
                            const int i = next.fetch_add(1);
                            if (i >= n_files) break;
                            const int fd = open((sub + "/f" + std::to_string(i)).c_str(), O_RDONLY);
                            if (fd < 0) continue;
                            if (at + fbytes + 4096 > cap) at = 0;
                            if (method == 4) posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
                            if (method == 0 || method == 4) {
                                size_t got = 0;
                                for (;;) { const ssize_t r = read(fd, buf + at + got, fbytes - got); if (r <= 0) break; got += (size_t)r; }
                            } else if (method == 3) {
                                size_t got = 0;
                                for (;;) {
                                    const ssize_t r = read(fd, bounce, 256u << 10);
                                    if (r <= 0) break;
                                    for (size_t o = 0; o < ((size_t)r & ~(size_t)31); o += 32)
                                        _mm256_stream_si256((__m256i*)(buf + at + got + o), _mm256_load_si256((const __m256i*)(bounce + o)));
                                    got += (size_t)r;
                                }
                                _mm_sfence();
                            } else {
                                void* m = mmap(nullptr, fbytes, PROT_READ, MAP_SHARED | (method == 2 ? MAP_POPULATE : 0), fd, 0);
                                if (method == 5) {
                                    for (size_t o = 0; o < (fbytes & ~(size_t)31); o += 32)
                                        _mm256_stream_si256((__m256i*)(buf + at + o), _mm256_loadu_si256((const __m256i*)((const char*)m + o)));
                                    _mm_sfence();
                                } else
                                memcpy(buf + at, m, fbytes);
                                munmap(m, fbytes);
                            }
                            at += (fbytes + 63) & ~(size_t)63;
                            close(fd);
                        }
                        free(bounce);
                        munmap(buf, cap);
                    });
                for (auto& t : th) t.join();
                printf(" pass %d %.1f GB/s", pass, n_files * (double)fbytes / (now() - t0) / 1e9);
            }
            printf("\n");
            fflush(stdout);
            for (int i = 0; i < n_files; ++i) unlink((sub + "/f" + std::to_string(i)).c_str());
            rmdir(sub.c_str());
        }
    return 0;
}
