// valu_rate.hip -- issue rate of the VALU ops the candidate kernel is made of (gfx950).
// hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITERS 2000
#define UNROLL 16   // independent chains per thread

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t r[UNROLL];
    float f[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { r[i] = seed + threadIdx.x * 7 + i; f[i] = (float)r[i]; }
    const uint32_t c = seed | 1;
    const float cf = 1.0001f;
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
            if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            if (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            if (OP == 4) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i]));
            if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r[i]) : "v"(c));
            if (OP == 6) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
            if (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
            if (OP == 8) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(r[i]));
            if (OP == 9) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(c));
            if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c));
            if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(uint64_t*)&f[i & ~1]) : "v"(*(uint64_t*)&f[(i & ~1)]));
            if (OP == 12) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            if (OP == 13) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            if (OP == 14) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
            if (OP == 15) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r[i]));
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += r[i] + (uint32_t)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, int blocks) {
    uint32_t* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(d, 3);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, 3);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * N_ITERS*UNROLL
    double winst = (double)blocks * 4 / 1024.0 * N_ITERS * UNROLL;
    printf("%-16s blocks=%5d  %8.3f ms  %.2f ns per wave-instr per SIMD (=%.2f cycles @2.4GHz)\n", name, blocks, ms,
           ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
    hipFree(d);
}

int main() {
    for (int blocks : {1024, 2048}) {   // 4 and 8 waves per SIMD
        run<0>("v_add_u32", blocks); run<13>("v_sub_u32", blocks); run<6>("v_add3_u32", blocks);
        run<3>("v_and_b32", blocks); run<15>("v_lshrrev_b32", blocks); run<8>("v_bfe_i32", blocks);
        run<2>("v_mul_u32_u24", blocks); run<14>("v_mad_u32_u24", blocks);
        run<5>("v_alignbit_b32", blocks); run<10>("v_cndmask_b32", blocks); run<9>("v_mov_dpp", blocks);
        run<4>("v_cvt_f32_u32", blocks); run<1>("v_fma_f32", blocks); run<7>("v_mul_f32", blocks);
        run<11>("v_pk_fma_f32", blocks); run<12>("v_pk_add_u16", blocks);
    }
    return 0;
}
