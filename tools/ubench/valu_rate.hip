// valu_rate.hip -- issue rate of the VALU ops the candidate kernel is made of (gfx950).
// hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
// Every op runs as UNROLL independent chains per lane, 4 and 8 waves per SIMD on every SIMD of the chip;
// the figure printed is nanoseconds (and cycles at 2.4 GHz) per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITERS 2000
#define UNROLL 16   // independent chains per thread

#define OPS(X) \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1") \
    X(1, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
    X(2, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1") \
    X(3, "v_and_b32", "v_and_b32 %0, %0, %1") \
    X(4, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0") \
    X(5, "v_bfe_i32", "v_bfe_i32 %0, %0, 3, 1") \
    X(6, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
    X(7, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %1") \
    X(8, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %1") \
    X(9, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 31") \
    X(10, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(11, "v_mov_dpp wave_shr", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    X(12, "v_mov_dpp row_shr", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    X(13, "v_add_u32_dpp wave_shr", "v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    X(14, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0") \
    X(15, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1") \
    X(16, "v_mul_f32", "v_mul_f32 %0, %0, %1") \
    X(17, "v_add_f32", "v_add_f32 %0, %0, %1") \
    X(18, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1") \
    X(19, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1") \
    X(20, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %1") \
    X(21, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1") \
    X(22, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %1") \
    X(23, "v_mad_u32_u16 op_sel", "v_mad_u32_u16 %0, %1, %1, %0 op_sel:[1,1,0,0]") \
    X(24, "v_dot2_u32_u16", "v_dot2_u32_u16 %0, %1, %1, %0") \
    X(25, "v_add_u32_sdwa", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
    X(26, "v_sub_u32_sdwa w0w0", "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0") \
    X(27, "v_min3_u32", "v_min3_u32 %0, %0, %1, %1") \
    X(28, "v_min_u32", "v_min_u32 %0, %0, %1") \
    X(29, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1") \
    X(30, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %1") \
    X(31, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %1") \
    X(32, "v_sqrt_f32", "v_sqrt_f32 %0, %0") \
    X(33, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
    X(34, "v_cmp_gt_f32 (vcc)", "v_cmp_gt_f32 vcc, %0, %1") \
    X(35, "v_cmp_gt_u32 (vcc)", "v_cmp_gt_u32 vcc, %0, %1") \
    X(36, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %1") \
    X(37, "v_sad_u32", "v_sad_u32 %0, %0, %1, %1") \
    X(38, "v_mqsad_pk_u16_u8", "v_add_u32 %0, %0, %1") \
    X(39, "v_max3_f32", "v_max3_f32 %0, %0, %1, %1")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t r[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r[i] = seed + threadIdx.x * 7 + i;
    const uint32_t c = seed | 1;
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
#define X(id, name, text) if (OP == id) asm volatile(text : "+v"(r[i]) : "v"(c) : "vcc");
            OPS(X)
#undef X
            if (OP == 100) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(uint64_t*)&r[i & ~1]) : "v"(*(uint64_t*)&r[(i & ~1)]));
            if (OP == 101) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(uint64_t*)&r[i & ~1]) : "v"(*(uint64_t*)&r[(i & ~1)]));
            if (OP == 102) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(uint64_t*)&r[i & ~1]) : "v"(*(uint64_t*)&r[(i & ~1)]));
            // a 50/50 mix of an integer and a float op (do they share the issue slot?)
            if (OP == 103) { if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c)); }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += r[i];
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
    printf("%-24s blocks=%5d  %8.3f ms  %.2f ns per wave-instr per SIMD (=%.2f cycles @2.4GHz)\n", name, blocks, ms,
           ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
    hipFree(d);
}

int main() {
    for (int blocks : {1024, 2048}) {   // 4 and 8 waves per SIMD
#define X(id, name, text) run<id>(name, blocks);
        OPS(X)
#undef X
        run<100>("v_pk_fma_f32", blocks); run<101>("v_pk_add_f32", blocks); run<102>("v_pk_mul_f32", blocks);
        run<103>("mix add_u32/fma_f32", blocks);
    }
    return 0;
}
