// Can a streaming kernel hand the machine to the NEXT one while its own last round of waves drains?  Two HIP streams take the
// kernels alternately; a kernel's last workgroup (dispatch is in order) writes a sequence number into signal memory as it starts,
// and the other stream waits for that value (hipStreamWaitValue32) before its kernel -- so kernel k + 1 flows into the slots kernel
// k's tail leaves, and the 12 us between two dispatches of ONE stream (profiles/r05h_region_anatomy.txt) never opens.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/wait_value tools/ubench/wait_value.hip && timeout -k 10 120 tools/ubench/wait_value
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64, 4) void k_work(uint32_t ticks, uint32_t* flag, uint32_t seq, uint32_t* sink) {
    __shared__ uint32_t lds[2304];   // 9 KB, as a streaming wave
    if (flag && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t t0 = wall_clock64();
    uint32_t it = 0;
    while (wall_clock64() - t0 < ticks && it < (1u << 22)) { __builtin_amdgcn_s_sleep(8); ++it; }
    if (it == 0xFFFFFFFFu) { lds[threadIdx.x] = it; *sink = lds[0]; }
}

int main(int argc, char** argv) {
    const int n_kernels = 20, waves = 15064;
    const uint32_t us = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 78;     // a wave's life; 15064 waves on 4096 slots: 3.68 rounds
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue: %d\n", can);
    if (!can) return 0;
    uint32_t *flag = nullptr, *sink = nullptr;
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&flag), 8, hipMallocSignalMemory));
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    CK(hipMemset(flag, 0, 8));
    hipStream_t st[2];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const uint32_t ticks = us * 100;   // wall_clock64: 100 MHz
    auto run = [&](bool two) -> float {
        uint32_t seq0 = 0;
        static uint32_t base = 0;
        base += 1000;
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(t0, st[0]);
        if (two) (void)hipStreamWaitEvent(st[1], t0, 0);
        for (int k = 0; k < n_kernels; ++k) {
            hipStream_t s = two ? st[k & 1] : st[0];
            if (two && k > 0) (void)hipStreamWaitValue32(s, flag, base + (uint32_t)k - 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
            hipLaunchKernelGGL(k_work, dim3(waves), dim3(64), 0, s, ticks, two ? flag : nullptr, base + (uint32_t)k, sink);
        }
        if (two) { hipEvent_t j; (void)hipEventCreate(&j); (void)hipEventRecord(j, st[1]); (void)hipStreamWaitEvent(st[0], j, 0); (void)hipEventDestroy(j); }
        (void)hipEventRecord(t1, st[0]);
        (void)hipEventSynchronize(t1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, t0, t1);
        (void)seq0;
        return ms;
    };
    for (int rep = 0; rep < 3; ++rep) {
        const float a = run(false), b = run(true);
        std::printf("%d kernels of %d waves x %u us: one stream %.3f ms (%.1f us per kernel), two streams + wait-value %.3f ms (%.1f us per kernel)\n",
                    n_kernels, waves, us, a, a * 1e3f / n_kernels, b, b * 1e3f / n_kernels);
    }
    if (hipGetLastError() != hipSuccess) std::printf("a HIP call failed\n");
    return 0;
}
