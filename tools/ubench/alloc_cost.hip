// alloc_cost.hip -- what the allocations of one ffs_stream cost on this box (hipMalloc / hipHostMalloc by size, alone and
// from several threads at once), and how fast a worker can copy a chunk from the page cache into pinned memory.
//   hipcc -O2 --offload-arch=gfx950 -o alloc_cost alloc_cost.hip -lpthread && ./alloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sched.h>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    for (size_t mb : {1, 16, 64, 256, 600}) {
        void* d = nullptr;
        auto t0 = std::chrono::steady_clock::now();
        (void)hipMalloc(&d, mb << 20);
        const double a = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        (void)hipMemset(d, 0, mb << 20);
        const double m = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        (void)hipFree(d);
        const double f = ms_since(t0);
        void* h = nullptr;
        t0 = std::chrono::steady_clock::now();
        (void)hipHostMalloc(&h, mb << 20, hipHostMallocDefault);
        const double ha = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        std::memset(h, 1, mb << 20);
        const double ht = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        (void)hipHostFree(h);
        const double hf = ms_since(t0);
        std::printf("%4zu MB: hipMalloc %.2f ms  memset %.2f  hipFree %.2f | hipHostMalloc %.2f ms  first touch %.2f  hipHostFree %.2f\n", mb, a, m, f, ha, ht, hf);
    }
    for (int nt : {1, 4, 16}) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        std::vector<void*> hp(nt, nullptr), dp(nt, nullptr);
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                (void)hipHostMalloc(&hp[t], (size_t)128 << 20, hipHostMallocDefault);
                (void)hipMalloc(&dp[t], (size_t)256 << 20);
            });
        for (auto& t : th) t.join();
        std::printf("%2d threads, each hipHostMalloc 128 MB + hipMalloc 256 MB: %.1f ms in all\n", nt, ms_since(t0));
        for (int t = 0; t < nt; ++t) { (void)hipHostFree(hp[t]); (void)hipFree(dp[t]); }
    }
    // page cache -> pinned: memcpy rate of one thread and of 8
    const size_t chunk = 8u << 20;
    std::vector<char> src(chunk, 3);
    void* pin = nullptr;
    (void)hipHostMalloc(&pin, chunk * 16, hipHostMallocDefault);
    std::memset(pin, 0, chunk * 16);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 64; ++i) std::memcpy((char*)pin + (i % 16) * chunk, src.data(), chunk);
    std::printf("memcpy into pinned, 1 thread: %.1f GB/s\n", 64.0 * chunk / ms_since(t0) / 1e6);
    t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 8; ++t)
            th.emplace_back([&, t] { for (int i = 0; i < 32; ++i) std::memcpy((char*)pin + (t * 2 + (i & 1)) * chunk, src.data(), chunk); });
        for (auto& t : th) t.join();
    }
    std::printf("memcpy into pinned, 8 threads: %.1f GB/s\n", 8 * 32.0 * chunk / ms_since(t0) / 1e6);
    // H2D of one 7.5 MB chunk alone, and 16 of them back to back
    void* d = nullptr;
    (void)hipMalloc(&d, chunk * 16);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
        t0 = std::chrono::steady_clock::now();
        (void)hipMemcpyAsync(d, pin, 7500000, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        const double one = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 16; ++i) (void)hipMemcpyAsync((char*)d + i * chunk, (char*)pin + i * chunk, 7500000, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        const double many = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        (void)hipMemcpyAsync(d, pin, chunk * 16, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        const double big = ms_since(t0);
        std::printf("H2D: one 7.5 MB chunk %.3f ms (%.1f GB/s); 16 chunks %.3f ms (%.1f GB/s); one 128 MB copy %.3f ms (%.1f GB/s)\n", one, 7.5 / one,
                    many, 16 * 7.5 / many, big, 16.0 * chunk / big / 1e6);
    }
    // H2D rate by the NUMA node the pinned buffer was allocated (and first touched) on: the GPU hangs off ONE socket
    {
        char bus[64] = {0};
        (void)hipDeviceGetPCIBusId(bus, sizeof bus, 0);
        for (char* q = bus; *q; ++q) *q = (char)tolower(*q);
        int gpu_node = -1;
        { std::ifstream f(std::string("/sys/bus/pci/devices/") + bus + "/numa_node"); f >> gpu_node; }
        std::printf("GPU 0 (%s) hangs off NUMA node %d\n", bus, gpu_node);
        cpu_set_t all;
        CPU_ZERO(&all);
        sched_getaffinity(0, sizeof all, &all);
        for (int node = 0; node < 8; ++node) {
            std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
            std::string list;
            if (!std::getline(f, list)) break;
            cpu_set_t set;
            CPU_ZERO(&set);
            std::stringstream ss(list);
            std::string tok;
            int n = 0;
            while (std::getline(ss, tok, ',')) {
                const size_t dash = tok.find('-');
                const int lo = atoi(tok.c_str()), hi = dash == std::string::npos ? lo : atoi(tok.c_str() + dash + 1);
                for (int c = lo; c <= hi; ++c) if (CPU_ISSET(c, &all)) { CPU_SET(c, &set); ++n; }
            }
            if (!n) { std::printf("node %d: no CPU of this process's affinity set\n", node); continue; }
            sched_setaffinity(0, sizeof set, &set);
            void* hp = nullptr;
            (void)hipHostMalloc(&hp, chunk * 16, hipHostMallocDefault);
            std::memset(hp, 5, chunk * 16);
            double best = 1e9, best1 = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                t0 = std::chrono::steady_clock::now();
                (void)hipMemcpyAsync(d, hp, chunk * 16, hipMemcpyHostToDevice, st);
                (void)hipStreamSynchronize(st);
                best = std::min(best, ms_since(t0));
                t0 = std::chrono::steady_clock::now();
                (void)hipMemcpyAsync(d, hp, 7500000, hipMemcpyHostToDevice, st);
                (void)hipStreamSynchronize(st);
                best1 = std::min(best1, ms_since(t0));
            }
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 16; ++i) std::memcpy((char*)hp + i * chunk, src.data(), chunk);
            const double cp = ms_since(t0);
            std::printf("pinned buffer from node %d (%d CPUs): H2D 128 MB %.1f GB/s, one 7.5 MB chunk %.3f ms (%.1f GB/s); memcpy into it %.1f GB/s\n", node, n,
                        16.0 * chunk / best / 1e6, best1, 7.5 / best1, 16.0 * chunk / cp / 1e6);
            (void)hipHostFree(hp);
        }
        sched_setaffinity(0, sizeof all, &all);
    }
    return 0;
}
