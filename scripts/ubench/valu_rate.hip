// Dev aid: relative issue cost of VALU instructions on gfx950 (one wave per SIMD, independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define BODY(INS) REP8(REP8(asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(c));))
#define KERNEL(name, INS)                                                              \
    __global__ __launch_bounds__(64) void name(uint32_t* out, int iters, uint32_t b, uint32_t c) { \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;              \
        for (int i = 0; i < iters; ++i) { BODY(INS) }                                  \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;                        \
    }
// each asm statement = 4 independent instructions
KERNEL(k_add_f32, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3")
KERNEL(k_cvt_pk, "v_cvt_pk_bf16_f32 %0, %0, %4\n v_cvt_pk_bf16_f32 %1, %1, %4\n v_cvt_pk_bf16_f32 %2, %2, %4\n v_cvt_pk_bf16_f32 %3, %3, %4")
KERNEL(k_xor, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4")
KERNEL(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %2, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %4, %5 bitop3:0x48\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x48\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x48\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x48")
KERNEL(k_perm, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %4, %0\n v_pk_sub_i16 %1, %4, %1\n v_pk_sub_i16 %2, %4, %2\n v_pk_sub_i16 %3, %4, %3")
KERNEL(k_pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %0\n v_pk_ashrrev_i16 %1, 15, %1\n v_pk_ashrrev_i16 %2, 15, %2\n v_pk_ashrrev_i16 %3, 15, %3")
// packed fp32: each instruction works on a register pair (2 values / lane)
#define KERNEL2(name, INS)                                                             \
    __global__ __launch_bounds__(64) void name(uint32_t* out, int iters, uint32_t b, uint32_t c) { \
        typedef float f2 __attribute__((ext_vector_type(2)));                          \
        f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, bb = {1.0001f, 0.9999f}; \
        for (int i = 0; i < iters; ++i) { REP8(REP8(asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(bb));)) } \
        out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(a0.x + a1.y + a2.x + a3.y);   \
    }
KERNEL2(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4")
KERNEL2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4")
KERNEL2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4")
KERNEL2(k_mov_b64, "v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4")
KERNEL(k_cmp_cnd, "v_cmp_le_u32 vcc, %5, %0\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_le_u32 vcc, %5, %2\n v_cndmask_b32 %3, 0, %3, vcc")
KERNEL(k_cmp_sdwa_cnd, "v_cmp_ge_u32_sdwa vcc, %0, %5 src0_sel:WORD_1 src1_sel:DWORD\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_ge_u32_sdwa vcc, %2, %5 src0_sel:WORD_1 src1_sel:DWORD\n v_cndmask_b32 %3, 0, %3, vcc")
KERNEL(k_max3, "v_max3_f32 %0, %0, %4, %1\n v_max3_f32 %1, %1, %4, %2\n v_max3_f32 %2, %2, %4, %3\n v_max3_f32 %3, %3, %4, %0")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 15, %0\n v_lshrrev_b32 %1, 15, %1\n v_lshrrev_b32 %2, 15, %2\n v_lshrrev_b32 %3, 15, %3")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4")

// 64-bit multiply-add: one instruction yields four 16-bit lots
__global__ __launch_bounds__(64) void k_mad_u64(uint32_t* out, int iters, uint32_t b, uint32_t c) {
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; ++i) {
        REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");))
    }
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3);
}

template <typename K>
float run(K k, const char* name, uint32_t* out, float base) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 4;   // one wave per SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 10, 3u, 5u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters, 3u, 5u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * 64 * 4);   // ns per instruction per wave
    printf("%-18s %8.3f ms  %6.3f ns/instr  x%.2f vs v_add_f32\n", name, ms, per, base > 0 ? per / base : 1.0);
    return (float)per;
}
int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 4 * 64 * 4);
    float base = run(k_add_f32, "v_add_f32", out, 0);
#define R(k) run(k, #k, out, base)
    R(k_mad_u64); R(k_fma_f32); R(k_mul_lo); R(k_mul_hi); R(k_mul_u24); R(k_mad_u24); R(k_exp); R(k_cvt_pk); R(k_xor); R(k_xor_sdwa); R(k_bitop3); R(k_perm);
    R(k_pk_sub_i16); R(k_pk_ashr_i16); R(k_pk_mul_f32); R(k_pk_add_f32); R(k_pk_fma_f32); R(k_mov_b64); R(k_cmp_cnd); R(k_cmp_sdwa_cnd); R(k_max3); R(k_lshr); R(k_and_or);
    return 0;
}
