// Dev aid: issue price of the vector instructions the attention loops are made of, measured as 12 independent fillers per
// v_mfma_f32_32x32x16_bf16 gap (well past the ~5 that hide under an MFMA, so the difference to 12 x v_fma_f32 is the
// instruction's price relative to a plain op), at one and at two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16x __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define R4(x) x x x x

enum { FMA, MAD64, PKSUB16, PKASHR16, AND32, MAX3, MULHI, MULLO, EXP, CNDVCC, CVTPK, PKADD32, PKMUL32, XOR3, LSHLOR, READ_U16, READ_B64, BFE, NKINDS };

template <int KIND>
__global__ __launch_bounds__(256, 2) void kern(float* out, int iters) {
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1; lds[threadIdx.x + 512] = 2; lds[threadIdx.x + 768] = 3;
    __syncthreads();
    f16x acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f + 0.25f); }
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, c = 1.0001f;
    uint32_t w0 = threadIdx.x * 2654435761u, w1 = w0 ^ 0x9e3779b9u, w2 = w0 + 77, w3 = w1 + 99, m = 0x7feb352du;
    uint64_t q0 = w0, q1 = w1, q2 = w2, q3 = w3;
    const uint32_t la = (threadIdx.x & 63) * 8;
    for (int it = 0; it < iters; ++it) {
#define MF(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
#define F3(TXT, ...) _Pragma("unroll") for (int k = 0; k < 3; ++k) asm volatile(TXT __VA_ARGS__);
#define FL()                                                                                                                       \
        if constexpr (KIND == FMA) { F3("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4", : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c)) } \
        else if constexpr (KIND == MAD64) { F3("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3", : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(w0), "v"(m) : "vcc") } \
        else if constexpr (KIND == PKSUB16) { F3("v_pk_sub_i16 %0, %0, %4 clamp\n v_pk_sub_i16 %1, %1, %4 clamp\n v_pk_sub_i16 %2, %2, %4 clamp\n v_pk_sub_i16 %3, %3, %4 clamp", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(m)) } \
        else if constexpr (KIND == PKASHR16) { F3("v_pk_ashrrev_i16 %0, 15, %0\n v_pk_ashrrev_i16 %1, 15, %1\n v_pk_ashrrev_i16 %2, 15, %2\n v_pk_ashrrev_i16 %3, 15, %3", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)) } \
        else if constexpr (KIND == AND32) { F3("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(m)) } \
        else if constexpr (KIND == MAX3) { F3("v_max3_f32 %0, %0, %4, %1\n v_max3_f32 %1, %1, %4, %2\n v_max3_f32 %2, %2, %4, %3\n v_max3_f32 %3, %3, %4, %0", : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c)) } \
        else if constexpr (KIND == MULHI) { F3("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(m)) } \
        else if constexpr (KIND == MULLO) { F3("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(m)) } \
        else if constexpr (KIND == EXP) { F3("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3", : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)) } \
        else if constexpr (KIND == CNDVCC) { F3("v_cmp_gt_i32 vcc, %4, %0\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_gt_i32 vcc, %4, %2\n v_cndmask_b32 %3, 0, %3, vcc", : "+v"(w0), "+v"(v1), "+v"(w2), "+v"(v3) : "v"(m) : "vcc") } \
        else if constexpr (KIND == CVTPK) { F3("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %1, %2, %3\n v_cvt_pk_bf16_f32 %2, %3, %0\n v_cvt_pk_bf16_f32 %3, %0, %1", : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)) } \
        else if constexpr (KIND == PKADD32) { F3("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2", : "+v"(q0), "+v"(q1) : "v"(q2)) } \
        else if constexpr (KIND == PKMUL32) { F3("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2", : "+v"(q0), "+v"(q1) : "v"(q2)) } \
        else if constexpr (KIND == XOR3) { F3("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(m)) } \
        else if constexpr (KIND == LSHLOR) { F3("v_lshl_or_b32 %0, %0, 3, %4\n v_lshl_or_b32 %1, %1, 3, %4\n v_lshl_or_b32 %2, %2, 3, %4\n v_lshl_or_b32 %3, %3, 3, %4", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(m)) } \
        else if constexpr (KIND == READ_U16) { F3("ds_read_i16 %0, %4\n ds_read_i16 %1, %4 offset:2\n ds_read_i16 %2, %4 offset:4\n ds_read_i16 %3, %4 offset:6\n s_waitcnt lgkmcnt(0)", : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(la)) } \
        else if constexpr (KIND == READ_B64) { F3("ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %1\n v_mov_b32 %1, %1\n v_mov_b32 %1, %1", : "=v"(q0), "+v"(w1) : "v"(la)) } \
        else if constexpr (KIND == BFE) { F3("v_bfe_i32 %0, %0, 0, 16\n v_bfe_i32 %1, %1, 0, 16\n v_bfe_i32 %2, %2, 0, 16\n v_bfe_i32 %3, %3, 0, 16", : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)) }
        R4(MF(acc0) FL() MF(acc1) FL() MF(acc2) FL() MF(acc3) FL())
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = v0 + v1 + v2 + v3 + (float)(w0 + w1 + w2 + w3) + (float)(q0 + q1 + q2 + q3);
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 65535] = s;
}

template <int KIND>
void run(float* out, const char* label) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float res[2];
    for (int wps = 1; wps <= 2; ++wps) {
        hipLaunchKernelGGL((kern<KIND>), dim3(256 * wps), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((kern<KIND>), dim3(256 * wps), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        res[wps - 1] = best * 1e6 / (iters * 16.0);      // ns per MFMA of one wave
    }
    printf("%-44s 1 wave/SIMD %7.2f ns per MFMA+12   2 waves/SIMD %7.2f ns (per wave; SIMD time per MFMA %6.2f)\n", label, res[0], res[1], res[1] / 2);
    fflush(stdout);
}

int main() {
    float* out; hipMalloc(&out, 65536 * 4);
    run<FMA>(out, "12 v_fma_f32");
    run<MAD64>(out, "12 v_mad_u64_u32");
    run<MULLO>(out, "12 v_mul_lo_u32");
    run<MULHI>(out, "12 v_mul_hi_u32");
    run<PKSUB16>(out, "12 v_pk_sub_i16 clamp");
    run<PKASHR16>(out, "12 v_pk_ashrrev_i16");
    run<AND32>(out, "12 v_and_b32");
    run<XOR3>(out, "12 v_xor_b32");
    run<LSHLOR>(out, "12 v_lshl_or_b32");
    run<BFE>(out, "12 v_bfe_i32");
    run<MAX3>(out, "12 v_max3_f32");
    run<EXP>(out, "12 v_exp_f32");
    run<CNDVCC>(out, "6 v_cmp (vcc) + 6 v_cndmask");
    run<CVTPK>(out, "12 v_cvt_pk_bf16_f32");
    run<PKADD32>(out, "12 v_pk_add_f32");
    run<PKMUL32>(out, "12 v_pk_mul_f32");
    run<READ_U16>(out, "12 ds_read_i16 (+3 waits)");
    run<READ_B64>(out, "3 ds_read_b64 + 9 v_mov");
    return 0;
}
