// The bubble between two dependent kernels of one HIP stream, seen from the device: every wave notes when it starts and ends
// (atomicMin / atomicMax of the 100 MHz wall clock); gap k = (first start of kernel k + 1) - (last end of kernel k).  By how the
// kernels are launched: plain, with start / stop events riding on the dispatch (hipExtLaunchKernelGGL: what the streaming kernels
// of the hot path do), with an event recorded behind each, and by the events' flags.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/launch_gap tools/ubench/launch_gap.hip && timeout -k 10 120 tools/ubench/launch_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_work(uint32_t ticks, unsigned long long* t_first, unsigned long long* t_last, uint32_t* sink) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(t_first, t0);
    uint32_t it = 0;
    while (wall_clock64() - t0 < ticks && it < (1u << 22)) { __builtin_amdgcn_s_sleep(4); ++it; }
    if (it == 0xFFFFFFFFu) *sink = it;
    if (threadIdx.x == 0) atomicMax(t_last, wall_clock64());
}

int main() {
    const int n = 40, waves = 4096 * 3;
    const uint32_t ticks = 50 * 100;   // 50 us per wave
    unsigned long long *d_first, *d_last;
    uint32_t* sink;
    CK(hipMalloc(&d_first, n * 8)); CK(hipMalloc(&d_last, n * 8)); CK(hipMalloc(&sink, 64));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<unsigned long long> first(n), last(n);
    auto report = [&](const char* what) {
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(first.data(), d_first, n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(last.data(), d_last, n * 8, hipMemcpyDeviceToHost);
        std::vector<double> gaps;
        for (int k = 5; k + 1 < n; ++k) gaps.push_back(((double)first[k + 1] - (double)last[k]) * 0.01);
        std::sort(gaps.begin(), gaps.end());
        double dur = 0;
        for (int k = 5; k < n; ++k) dur += ((double)last[k] - (double)first[k]) * 0.01;
        std::printf("%-64s gap between dependent kernels: median %5.1f us (min %5.1f, max %5.1f); kernel %6.1f us\n", what, gaps[gaps.size() / 2], gaps.front(), gaps.back(), dur / (n - 5));
    };
    auto reset = [&]() { (void)hipMemset(d_first, 0xFF, n * 8); (void)hipMemset(d_last, 0, n * 8); (void)hipDeviceSynchronize(); };
    for (int rep = 0; rep < 2; ++rep) {
        reset();
        for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_work, dim3(waves), dim3(64), 0, st, ticks, d_first + k, d_last + k, sink);
        report("plain launches");
        for (unsigned flags : {0u, (unsigned)hipEventDisableTiming, (unsigned)hipEventReleaseToDevice, (unsigned)hipEventReleaseToSystem, (unsigned)hipEventDisableSystemFence}) {
            std::vector<hipEvent_t> ev(2 * n);
            bool ok = true;
            for (auto& e : ev) ok = ok && hipEventCreateWithFlags(&e, flags) == hipSuccess;
            if (!ok) { std::printf("events with flags %#x: not created\n", flags); (void)hipGetLastError(); continue; }
            char what[96];
            reset();
            for (int k = 0; k < n; ++k) hipExtLaunchKernelGGL(k_work, dim3(waves), dim3(64), 0, st, ev[2 * k], ev[2 * k + 1], 0, ticks, d_first + k, d_last + k, sink);
            std::snprintf(what, sizeof what, "start / stop events on the dispatch, event flags %#x", flags);
            report(what);
            reset();
            for (int k = 0; k < n; ++k) {
                hipLaunchKernelGGL(k_work, dim3(waves), dim3(64), 0, st, ticks, d_first + k, d_last + k, sink);
                (void)hipEventRecord(ev[2 * k + 1], st);
            }
            std::snprintf(what, sizeof what, "plain launch + hipEventRecord behind it, event flags %#x", flags);
            report(what);
            for (auto& e : ev) (void)hipEventDestroy(e);
        }
    }
    return 0;
}
