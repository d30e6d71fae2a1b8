// Dev aid: cost per v_mfma_f32_32x32x16_bf16 of one wave per SIMD as a function of the register form of the MFMA
// (VGPR / AGPR accumulator, AGPR B operand), of an s_nop 1 prefix, and of k independent vector fillers per MFMA gap
// (plain v_fma_f32, v_exp_f32, v_pk_fma_f32, v_cmp + v_cndmask through an SGPR pair, v_cvt_pk_bf16_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16x __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define R4(x) x x x x
#define R16(x) R4(R4(x))

// FORM 0: acc VGPR, A/B VGPR   1: acc VGPR, B AGPR   2: acc AGPR, A/B VGPR   3: as 2 with "s_nop 1" in front
// FILL 0: none  1: k x v_fma_f32  2: k/4 x (2 v_exp + 2 v_fma)  3: k/2 x v_pk_fma_f32 (same flops as k fma)  4: k/4 x (2 v_cmp_e64 + 2 v_cndmask_e64)
//      5: k x v_cvt_pk_bf16_f32
template <int FORM, int FILL, int K>
__global__ __launch_bounds__(256, 1) void kern(float* out, int iters) {
    f16x acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f + 0.25f); }
    bf8 ba = b;
    if constexpr (FORM == 1) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(*(float*)&ba) : "v"(*(float*)&b));   // placeholder (lane 0 dword only matters for timing)
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, c = 1.0001f;
    f2 p0 = {v0, v1}, p1 = {v2, v3}, pc = {c, c};
    uint32_t w0 = threadIdx.x * 2654435761u, w1 = w0 ^ 0x9e3779b9u;
    for (int it = 0; it < iters; ++it) {
#define MF(ACC)                                                                                                            \
        if constexpr (FORM == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));          \
        else if constexpr (FORM == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "a"(ba));    \
        else if constexpr (FORM == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b));     \
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b));
#define FL()                                                                                                               \
        if constexpr (FILL == 1) { _Pragma("unroll") for (int k = 0; k < K / 4; ++k)                                        \
            asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c)); }                                             \
        else if constexpr (FILL == 2) { _Pragma("unroll") for (int k = 0; k < K / 4; ++k)                                   \
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"        \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c)); }                                             \
        else if constexpr (FILL == 3) { _Pragma("unroll") for (int k = 0; k < K / 4; ++k)                                   \
            asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(p0), "+v"(p1) : "v"(pc)); }     \
        else if constexpr (FILL == 4) { _Pragma("unroll") for (int k = 0; k < K / 4; ++k) { unsigned long long m0, m1;      \
            asm volatile("v_cmp_ge_i32_e64 %2, %4, %5\n v_cmp_ge_i16_e64 %3, %5, %4\n v_cndmask_b32_e64 %0, 0, %0, %2\n v_cndmask_b32_e64 %1, 0, %1, %3" \
                         : "+v"(v0), "+v"(v1), "=&s"(m0), "=&s"(m1) : "v"(w0), "v"(w1)); } }                                  \
        else if constexpr (FILL == 5) { _Pragma("unroll") for (int k = 0; k < K / 4; ++k)                                   \
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %1, %2, %3\n v_cvt_pk_bf16_f32 %2, %3, %0\n v_cvt_pk_bf16_f32 %3, %0, %1" \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)); }
        R4(MF(acc0) FL() MF(acc1) FL() MF(acc2) FL() MF(acc3) FL())
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = v0 + v1 + v2 + v3 + p0[0] + p0[1] + p1[0] + p1[1];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM, int FILL, int K>
void run(float* out, const char* label) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<FORM, FILL, K>), dim3(256), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern<FORM, FILL, K>), dim3(256), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-58s %7.2f ns / MFMA  (%.3f ms)\n", label, best * 1e6 / (iters * 16.0), best);
    fflush(stdout);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<0, 0, 0>(out, "acc VGPR, A/B VGPR, no filler");
    run<1, 0, 0>(out, "acc VGPR, B AGPR, no filler");
    run<2, 0, 0>(out, "acc AGPR, no filler");
    run<3, 0, 0>(out, "acc AGPR, s_nop 1 prefix, no filler");
    run<0, 1, 4>(out, "acc VGPR + 4 v_fma");
    run<0, 1, 8>(out, "acc VGPR + 8 v_fma");
    run<0, 1, 12>(out, "acc VGPR + 12 v_fma");
    run<1, 1, 8>(out, "acc VGPR, B AGPR + 8 v_fma");
    run<2, 1, 4>(out, "acc AGPR + 4 v_fma");
    run<2, 1, 8>(out, "acc AGPR + 8 v_fma");
    run<2, 1, 12>(out, "acc AGPR + 12 v_fma");
    run<3, 1, 8>(out, "acc AGPR, s_nop 1 + 8 v_fma");
    run<2, 2, 8>(out, "acc AGPR + 4 v_exp + 4 v_fma");
    run<0, 2, 8>(out, "acc VGPR + 4 v_exp + 4 v_fma");
    run<2, 3, 8>(out, "acc AGPR + 4 v_pk_fma (= 8 fma of work)");
    run<0, 3, 8>(out, "acc VGPR + 4 v_pk_fma (= 8 fma of work)");
    run<2, 4, 8>(out, "acc AGPR + 4 v_cmp_e64 + 4 v_cndmask_e64");
    run<0, 4, 8>(out, "acc VGPR + 4 v_cmp_e64 + 4 v_cndmask_e64");
    run<2, 5, 8>(out, "acc AGPR + 8 v_cvt_pk_bf16_f32");
    run<0, 5, 8>(out, "acc VGPR + 8 v_cvt_pk_bf16_f32");
    return 0;
}
