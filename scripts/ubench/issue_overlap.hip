// Dev aid: how VALU and MFMA issue share a gfx950 SIMD.
//   * shader clock under load (s_memtime ticks per wall_clock64 tick)
//   * VALU and MFMA throughput with 1 / 2 / 3 waves per SIMD
//   * VALU waves and MFMA waves resident on the same SIMD: does their work overlap?
//   * one wave alternating 1 MFMA with k independent VALU instructions: how many fit in the MFMA shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16x __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)

struct Clk { uint64_t shader, wall; };

// mode bit layout: waves [0, nm) of each SIMD's share run MFMA chains, the rest VALU chains.
template <int VALU_PER_MFMA>
__global__ __launch_bounds__(1024) void mixed_same_wave(float* out, int iters) {
    f16x acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, c = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#define STEP(ACC)                                                                           \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, ACC, 0, 0, 0);                  \
        _Pragma("unroll") for (int k = 0; k < VALU_PER_MFMA / 4; ++k)                       \
            asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c));
        REP4(STEP(acc0) STEP(acc1) STEP(acc2) STEP(acc3))
#undef STEP
    }
    float s = v0 + v1 + v2 + v3;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same split, 32x32x16 with the accumulators pinned to AGPRs
__global__ __launch_bounds__(1024) void split_wavesA(float* out, int iters, int wm, int valu_scale, Clk* clk) {
    const int wave = threadIdx.x >> 6;
    const bool is_mfma = (wave >> 2) < wm;
    float s = 0;
    if (is_mfma) {
        f16x acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        bf8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        for (int it = 0; it < iters; ++it) {
            REP4(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n"
                              "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                              : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3) : "v"(a), "v"(b));)
        }
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    } else {
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, c = 1.0001f;
        for (int it = 0; it < iters * valu_scale; ++it) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c));)
        }
        s = v0 + v1 + v2 + v3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f4x __attribute__((ext_vector_type(4)));
// same split, MFMA waves run 16x16x32 bf16 (two per 32x32x16-equivalent of work)
__global__ __launch_bounds__(1024) void split_waves16(float* out, int iters, int wm, int valu_scale, Clk* clk) {
    const int wave = threadIdx.x >> 6;
    const bool is_mfma = (wave >> 2) < wm;
    float s = 0;
    if (is_mfma) {
        f4x acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        bf8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        for (int it = 0; it < iters; ++it) {
            REP8(acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
                 acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);)
        }
        for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    } else {
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, c = 1.0001f;
        for (int it = 0; it < iters * valu_scale; ++it) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c));)
        }
        s = v0 + v1 + v2 + v3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves_mfma / waves_valu per SIMD (block = 4 * (wm + wv) waves; wave w sits on SIMD w % 4)
__global__ __launch_bounds__(1024) void split_waves(float* out, int iters, int wm, int valu_scale, Clk* clk) {
    const int wave = threadIdx.x >> 6;
    const bool is_mfma = (wave >> 2) < wm;
    uint64_t t0 = 0, w0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    float s = 0;
    if (is_mfma) {
        f16x acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        bf8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        for (int it = 0; it < iters; ++it) {
            REP4(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
                 acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);)
        }
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    } else {
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, c = 1.0001f;
        for (int it = 0; it < iters * valu_scale; ++it) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c));)
        }
        s = v0 + v1 + v2 + v3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk->shader = __builtin_readcyclecounter() - t0;
        clk->wall = wall_clock64() - w0;
    }
}

static float timed(void (*launch)(void*), void* ctx) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(ctx); hipDeviceSynchronize();
    hipEventRecord(e0); launch(ctx); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; Clk* clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, sizeof(Clk));
    const int iters = 4000;
    printf("# block = 4*(wm+wv) waves on one CU (256 blocks); MFMA waves run %d x 16 MFMA 32x32x16 bf16, VALU waves %d x 64 x scale v_fma_f32\n", iters, iters);
    struct Ctx { float* out; int iters, wm, wv, scale; Clk* clk; } c;
    auto go = [](void* p) {
        Ctx* c = (Ctx*)p;
        hipLaunchKernelGGL(split_waves, dim3(256), dim3(256 * (c->wm + c->wv)), 0, 0, c->out, c->iters, c->wm, c->scale, c->clk);
    };
    const int cfgs[][3] = {{1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {0, 1, 2}, {0, 2, 2}, {0, 3, 2}, {0, 4, 2}, {1, 1, 2}, {1, 1, 1}, {1, 2, 1}, {2, 2, 2}, {1, 1, 4}, {1, 3, 1}};
    for (auto& k : cfgs) {
        c = {out, iters, k[0], k[1], k[2], clk};
        float ms = timed(go, &c);
        Clk h; hipMemcpy(&h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double n_mfma = (double)iters * 16 * k[0], n_valu = (double)iters * 64 * k[2] * k[1];
        printf("mfma waves/SIMD %d  valu waves/SIMD %d (scale %d): %8.3f ms | per SIMD: %7.2f ns/MFMA  %6.3f ns/VALU | shader clock %.3f GHz\n",
               k[0], k[1], k[2], ms, n_mfma > 0 ? ms * 1e6 / n_mfma : 0.0, n_valu > 0 ? ms * 1e6 / n_valu : 0.0,
               (double)h.shader / ((double)h.wall * 10.0));   // wall_clock64 ticks at 100 MHz
    }
    printf("# 16x16x32 bf16 MFMA waves (%d x 32 MFMA = same FLOPs), same split\n", iters);
    auto go16 = [](void* p) {
        Ctx* c = (Ctx*)p;
        hipLaunchKernelGGL(split_waves16, dim3(256), dim3(256 * (c->wm + c->wv)), 0, 0, c->out, c->iters, c->wm, c->scale, c->clk);
    };
    const int cfgs16[][3] = {{1, 0, 0}, {2, 0, 0}, {1, 1, 1}, {1, 1, 2}, {1, 2, 1}, {1, 3, 1}, {2, 2, 2}};
    for (auto& k : cfgs16) {
        c = {out, iters, k[0], k[1], k[2], clk};
        float ms = timed(go16, &c);
        const double n_mfma = (double)iters * 32 * k[0], n_valu = (double)iters * 64 * k[2] * k[1];
        printf("mfma16 waves/SIMD %d  valu waves/SIMD %d (scale %d): %8.3f ms | per SIMD: %7.2f ns/MFMA  %6.3f ns/VALU\n",
               k[0], k[1], k[2], ms, n_mfma > 0 ? ms * 1e6 / n_mfma : 0.0, n_valu > 0 ? ms * 1e6 / n_valu : 0.0);
    }
    printf("# 32x32x16 with AGPR accumulators, same split\n");
    auto goA = [](void* p) {
        Ctx* c = (Ctx*)p;
        hipLaunchKernelGGL(split_wavesA, dim3(256), dim3(256 * (c->wm + c->wv)), 0, 0, c->out, c->iters, c->wm, c->scale, c->clk);
    };
    for (auto& k : cfgs16) {
        c = {out, iters, k[0], k[1], k[2], clk};
        float ms = timed(goA, &c);
        const double n_mfma = (double)iters * 16 * k[0], n_valu = (double)iters * 64 * k[2] * k[1];
        printf("mfmaA waves/SIMD %d  valu waves/SIMD %d (scale %d): %8.3f ms | per SIMD: %7.2f ns/MFMA  %6.3f ns/VALU\n",
               k[0], k[1], k[2], ms, n_mfma > 0 ? ms * 1e6 / n_mfma : 0.0, n_valu > 0 ? ms * 1e6 / n_valu : 0.0);
    }
    printf("# one wave per SIMD, each MFMA followed by k independent v_fma_f32\n");
#define MIX(K) { struct C2 { float* o; int it; } c2 = {out, iters};                                                        \
        float ms = timed([](void* p) { C2* c = (C2*)p; hipLaunchKernelGGL(mixed_same_wave<K>, dim3(256), dim3(256), 0, 0, c->o, c->it); }, &c2); \
        printf("k = %2d: %8.3f ms  %7.2f ns per (MFMA + k VALU)\n", K, ms, ms * 1e6 / ((double)iters * 16)); }
    MIX(0) MIX(4) MIX(8) MIX(12) MIX(16) MIX(24)
    printf("# two waves per SIMD, same kernel\n");
#define MIX2(K) { struct C2 { float* o; int it; } c2 = {out, iters};                                                       \
        float ms = timed([](void* p) { C2* c = (C2*)p; hipLaunchKernelGGL(mixed_same_wave<K>, dim3(256), dim3(512), 0, 0, c->o, c->it); }, &c2); \
        printf("k = %2d: %8.3f ms  %7.2f ns per SIMD per (MFMA + k VALU)\n", K, ms, ms * 1e6 / ((double)iters * 32)); }
    MIX2(0) MIX2(4) MIX2(8) MIX2(12) MIX2(16) MIX2(24)
    return 0;
}
