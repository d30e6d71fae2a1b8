// LayerNorm (+ AdaLN modulate) forward / backward for gfx950.  HBM-bound: one wave per token row,
// 16-byte loads, wavefront (64-lane) shuffle reductions, one pass over x in each direction.
//
// Reference semantics:
//   models/hybrid_vit_backbone.py:84-86,229   nn.LayerNorm(C), eps 1e-5, affine
//   models/hybrid_vit_backbone.py:120-121     x_mod = (1 + scale_sa) * norm1(x) + shift_sa
//   models/hybrid_vit_backbone.py:136-137     x_mod2 = (1 + scale_mlp) * norm3(x) + shift_mlp
// The residual stream x is fp32; y is written in the dtype the following GEMM consumes.
//
// Backward produces dx (optionally summed with the gradient already on the residual stream),
// and two-stage deterministic column reductions for dgamma, dbeta and the per-sample dscale, dshift.
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

constexpr int kMaxPerLane = 16;   // C <= 1024

// Lane-local view of one row: VEC: 4 float4 at columns 4*(lane + 64 t) ; scalar: 16 floats at lane + 64 t.
template <bool VEC>
struct RowMap {
    static __device__ __forceinline__ int col(int e, int lane) {
        if constexpr (VEC) return 4 * (lane + 64 * (e >> 2)) + (e & 3);
        else return lane + 64 * e;
    }
};

template <bool VEC, int PL>
__device__ __forceinline__ void load_row_f32(const float* p, int C, int lane, float (&v)[PL]) {
    if constexpr (VEC) {
#pragma unroll
        for (int t = 0; t < PL / 4; ++t) {
            int c = 4 * (lane + 64 * t);
            f32x4 x = c < C ? *reinterpret_cast<const f32x4*>(p + c) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * t + j] = x[j];
        }
    } else {
#pragma unroll
        for (int e = 0; e < PL; ++e) { int c = lane + 64 * e; v[e] = c < C ? p[c] : 0.f; }
    }
}

template <typename T, bool VEC, int PL>
__device__ __forceinline__ void load_row(const T* p, int C, int lane, float (&v)[PL]) {
    if constexpr (sizeof(T) == 4) { load_row_f32<VEC, PL>(reinterpret_cast<const float*>(p), C, lane, v); }
    else {
        if constexpr (VEC) {
#pragma unroll
            for (int t = 0; t < PL / 4; ++t) {
                int c = 4 * (lane + 64 * t);
                bf16x4 x = c < C ? *reinterpret_cast<const bf16x4*>(p + c) : bf16x4{0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * t + j] = bf2f(x[j]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < PL; ++e) { int c = lane + 64 * e; v[e] = c < C ? bf2f(p[c]) : 0.f; }
        }
    }
}

template <typename T, bool VEC, int PL>
__device__ __forceinline__ void store_row(T* p, int C, int lane, const float (&v)[PL]) {
    if constexpr (VEC) {
#pragma unroll
        for (int t = 0; t < PL / 4; ++t) {
            int c = 4 * (lane + 64 * t);
            if (c < C) {
                if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(p + c) = f32x4{v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]};
                else *reinterpret_cast<bf16x4*>(p + c) = bf16x4{f2bf(v[4 * t]), f2bf(v[4 * t + 1]), f2bf(v[4 * t + 2]), f2bf(v[4 * t + 3])};
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < PL; ++e) { int c = lane + 64 * e; if (c < C) p[c] = from_f<T>(v[e]); }
    }
}

template <typename TO, bool VEC, int PL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    float gam[PL], bet[PL];
    load_row_f32<VEC, PL>(a.gamma, C, lane, gam);
    load_row_f32<VEC, PL>(a.beta, C, lane, bet);
    const float invC = 1.f / (float)C;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < a.rows; row += (int64_t)gridDim.x * 4) {
        float x[PL];
        load_row_f32<VEC, PL>(a.x + row * C, C, lane, x);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < PL; ++e) s += x[e];
        const float mean = wave_sum(s) * invC;
        float vs = 0.f;
#pragma unroll
        for (int e = 0; e < PL; ++e) {
            float d = RowMap<VEC>::col(e, lane) < C ? x[e] - mean : 0.f;
            vs += d * d;
        }
        const float rstd = rsqrtf(wave_sum(vs) * invC + a.eps);
        float y[PL];
#pragma unroll
        for (int e = 0; e < PL; ++e) y[e] = (x[e] - mean) * rstd * gam[e] + bet[e];
        if (a.scale) {
            const int64_t bidx = row / a.rows_per_batch;
            float sc[PL], sh[PL];
            load_row_f32<VEC, PL>(a.scale + bidx * C, C, lane, sc);
            load_row_f32<VEC, PL>(a.shift + bidx * C, C, lane, sh);
#pragma unroll
            for (int e = 0; e < PL; ++e) y[e] = y[e] * (1.f + sc[e]) + sh[e];
        }
        store_row<TO, VEC, PL>(reinterpret_cast<TO*>(a.y) + row * C, C, lane, y);
        if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
    }
}

// grid = nbatch * blocks_per_batch ; block handles a contiguous slice of one sample's rows.
template <typename TO, bool VEC, int PL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][4 kinds][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    const int bpb = a.blocks_per_batch;
    const int bidx = blockIdx.x / bpb, blk = blockIdx.x % bpb;
    const int rpb = a.rows_per_batch;
    const int per = (rpb + bpb - 1) / bpb;
    const int r0 = blk * per, r1 = min(rpb, r0 + per);
    float gam[PL], bet[PL], sc[PL];
    load_row_f32<VEC, PL>(a.gamma, C, lane, gam);
    load_row_f32<VEC, PL>(a.beta, C, lane, bet);
    const bool mod = a.scale != nullptr;
    if (mod) load_row_f32<VEC, PL>(a.scale + (int64_t)bidx * C, C, lane, sc);
    float dg[PL], db[PL], dsc[PL], dsh[PL];
#pragma unroll
    for (int e = 0; e < PL; ++e) { dg[e] = 0.f; db[e] = 0.f; dsc[e] = 0.f; dsh[e] = 0.f; }
    const float invC = 1.f / (float)C;
    // software pipeline: the loads of this wave's next row are issued before the current row's reductions
    float xn[PL], dyn[PL], drn[PL];
    float mean_n = 0.f, rstd_n = 0.f;
    auto fetch = [&](int rr) {
        const int64_t row = (int64_t)bidx * rpb + rr;
        load_row_f32<VEC, PL>(a.x + row * C, C, lane, xn);
        load_row<TO, VEC, PL>(reinterpret_cast<const TO*>(a.dy) + row * C, C, lane, dyn);
        if (a.dres) load_row_f32<VEC, PL>(a.dres + row * C, C, lane, drn);
        mean_n = a.mean[row];
        rstd_n = a.rstd[row];
    };
    if (r0 + wave < r1) fetch(r0 + wave);
    for (int rr = r0 + wave; rr < r1; rr += 4) {
        const int64_t row = (int64_t)bidx * rpb + rr;
        float x[PL], dy[PL], dx[PL];
#pragma unroll
        for (int e = 0; e < PL; ++e) { x[e] = xn[e]; dy[e] = dyn[e]; dx[e] = a.dres ? drn[e] : 0.f; }
        const float mean = mean_n, rstd = rstd_n;
        if (rr + 4 < r1) fetch(rr + 4);
        float g[PL], xh[PL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < PL; ++e) {
            const bool in = RowMap<VEC>::col(e, lane) < C;
            xh[e] = in ? (x[e] - mean) * rstd : 0.f;
            const float dyln = mod ? dy[e] * (1.f + sc[e]) : dy[e];
            if (mod) { dsc[e] += dy[e] * (xh[e] * gam[e] + bet[e]); dsh[e] += dy[e]; }
            dg[e] += dyln * xh[e];
            db[e] += dyln;
            g[e] = dyln * gam[e];
            s1 += g[e];
            s2 += g[e] * xh[e];
        }
        const float c1 = wave_sum(s1) * invC, c2 = wave_sum(s2) * invC;
#pragma unroll
        for (int e = 0; e < PL; ++e) dx[e] += rstd * (g[e] - c1 - xh[e] * c2);
        store_row<float, VEC, PL>(a.dx + row * C, C, lane, dx);
    }
    // cross-wave reduction in a fixed order (deterministic)
    float* mine = red + (size_t)wave * 4 * C;
#pragma unroll
    for (int e = 0; e < PL; ++e) {
        const int c = RowMap<VEC>::col(e, lane);
        if (c < C) { mine[c] = dg[e]; mine[C + c] = db[e]; mine[2 * C + c] = dsc[e]; mine[3 * C + c] = dsh[e]; }
    }
    __syncthreads();
    float* out = a.partial + (size_t)blockIdx.x * 4 * C;
    for (int i = threadIdx.x; i < 4 * C; i += 256)
        out[i] = ((red[i] + red[4 * C + i]) + red[8 * C + i]) + red[12 * C + i];
}

// grid (ceil(C/8), 2 + 2*nbatch): y = 0 dgamma, 1 dbeta (over all blocks); y = 2 + 2b + k: dscale / dshift of sample b.
__global__ __launch_bounds__(256) void ln_bwd_final_kernel(const LnArgs a, int nbatch) {
    __shared__ float red[32][kFinalCols];
    const int C = a.C, bpb = a.blocks_per_batch;
    const int c = kFinalCols * blockIdx.x + (threadIdx.x & 7);
    const bool valid = c < C;
    const int y = blockIdx.y;
    float* dst;
    float s;
    if (y < 2) {
        s = block_colsum8(a.partial + (size_t)y * C, nbatch * bpb, 4 * (int64_t)C, c, valid, red);
        dst = (y == 0 ? a.dgamma : a.dbeta) + c;
    } else {
        const int b = (y - 2) >> 1, k = (y - 2) & 1;
        s = block_colsum8(a.partial + ((size_t)b * bpb * 4 + 2 + k) * C, bpb, 4 * (int64_t)C, c, valid, red);
        dst = (k == 0 ? a.dscale : a.dshift) + (size_t)b * C + c;
    }
    if (valid && (threadIdx.x >> 3) == 0) *dst = s;
}

}  // namespace

int layernorm_bwd_blocks_per_batch(int rows_per_batch) {
    // One wave walks its rows one after another with the next row's loads in flight; 32 rows per block (8 per wave)
    // up to 256 blocks per sample: beyond that the per-block epilogue (LDS fold + 4 x C partial floats) and the final
    // reduction outweigh the extra parallelism (measured: 2048 blocks at 65536 rows 88 us vs 68 us at 512).
    int b = (rows_per_batch + 31) / 32;
    if (b > 256) b = 256;
    if (b < 1) b = 1;
    return b;
}

hipError_t layernorm_fwd_launch(const LnArgs& a, hipStream_t st) {
    if (a.C > 64 * kMaxPerLane || a.C < 1) return hipErrorInvalidValue;
    const bool vec = (a.C % 4) == 0;
    int64_t blocks = (a.rows + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), blk(256);
    // rows up to 256 wide (the models' C) keep 4 values per lane; wider rows (<= 1024) 16
#define HVC_LN_FWD(TO, VEC) \
    do { if (a.C <= 256) hipLaunchKernelGGL((ln_fwd_kernel<TO, VEC, 4>), grid, blk, 0, st, a); \
         else hipLaunchKernelGGL((ln_fwd_kernel<TO, VEC, kMaxPerLane>), grid, blk, 0, st, a); } while (0)
    if (a.out_bf16) {
        if (vec) HVC_LN_FWD(bf16, true); else HVC_LN_FWD(bf16, false);
    } else {
        if (vec) HVC_LN_FWD(float, true); else HVC_LN_FWD(float, false);
    }
#undef HVC_LN_FWD
    return hipGetLastError();
}

hipError_t layernorm_bwd_launch(const LnArgs& a, hipStream_t st) {
    if (a.C > 64 * kMaxPerLane || a.C < 1 || a.rows % a.rows_per_batch != 0) return hipErrorInvalidValue;
    const bool vec = (a.C % 4) == 0;
    const int nbatch = a.rows / a.rows_per_batch;
    dim3 grid((unsigned)(nbatch * a.blocks_per_batch)), blk(256);
    const size_t lds = (size_t)16 * a.C * sizeof(float);
#define HVC_LN_BWD(TO, VEC) \
    do { if (a.C <= 256) hipLaunchKernelGGL((ln_bwd_kernel<TO, VEC, 4>), grid, blk, lds, st, a); \
         else hipLaunchKernelGGL((ln_bwd_kernel<TO, VEC, kMaxPerLane>), grid, blk, lds, st, a); } while (0)
    if (a.out_bf16) {
        if (vec) HVC_LN_BWD(bf16, true); else HVC_LN_BWD(bf16, false);
    } else {
        if (vec) HVC_LN_BWD(float, true); else HVC_LN_BWD(float, false);
    }
#undef HVC_LN_BWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ln_bwd_final_kernel, dim3((a.C + kFinalCols - 1) / kFinalCols, 2 + (a.dscale ? 2 * nbatch : 0)), blk, 0, st, a, nbatch);
    return hipGetLastError();
}

}  // namespace hvc
