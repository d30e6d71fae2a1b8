// DRR ray-sum projections for gfx950 (HBM-bound: one read of the volume forward, one read + one
// write backward).
//
// Reference semantics:
//   models/diagnostic_losses.py:45-63  DRRRenderer: att = exp(-0.3 (v + 1)); angle 90: att.sum(-1)
//        -> (B,D,H) -> transpose -> (B,H,D); else att.sum(1) -> (B,H,W); clamp(min=1e-6)
//   direct_regression/progressive_cascade/loss_multiscale.py:260-267  DRRReprojectionLoss.generate_drr:
//        mean over D -> (B,1,H,W)   /  mean over W -> (B,1,D,H)   (no exp, no transpose, no clamp)
//
// axis 0 (along D): lanes run along the contiguous H*W plane (16-byte loads), the four waves of a
//   workgroup take interleaved depth slices and combine through LDS.
// axis 2 (along W): a row of W voxels is spread over a group of lanes (16-byte loads) and summed
//   with wavefront shuffles (DPP/ds_swizzle row reductions) -- no LDS, no atomics.
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };
template <> struct Vec4<bf16> { typedef bf16x4 type; };

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]) {
    typename Vec4<T>::type x = *reinterpret_cast<const typename Vec4<T>::type*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = to_f<T>(x[j]);
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]) {
    typename Vec4<T>::type x;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = from_f<T>(v[j]);
    *reinterpret_cast<typename Vec4<T>::type*>(p) = x;
}

__device__ __forceinline__ float drr_f(float v, int exp_mode, float mu) {
    return exp_mode ? __expf(-mu * (v + 1.f)) : v;
}

// ---- axis 0 : out[b][hw] = clamp(scale * sum_d f(vol[b][d][hw])) ; VEC4 over hw -----------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void drr_fwd_d_kernel(const DrrArgs a) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t HW = (int64_t)a.H * a.W;
    constexpr int E = VEC ? 4 : 1;
    const int64_t per_b = (HW + 64 * E - 1) / (64 * E);      // blocks per sample
    const int b = (int)(blockIdx.x / per_b);
    const int64_t hw = ((int64_t)(blockIdx.x % per_b) * 64 + lane) * E;
    const T* vp = reinterpret_cast<const T*>(a.vol) + (int64_t)b * a.D * HW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (hw < HW) {
        for (int d = wave; d < a.D; d += 4) {
            if constexpr (VEC) {
                float v[4];
                load4<T>(vp + (int64_t)d * HW + hw, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += drr_f(v[j], a.exp_mode, a.mu);
            } else {
                acc[0] += drr_f(to_f<T>(vp[(int64_t)d * HW + hw]), a.exp_mode, a.mu);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) red[wave][lane * E + j] = acc[j];
    __syncthreads();
    if (wave == 0 && hw < HW) {
        T* op = reinterpret_cast<T*>(a.out) + (int64_t)b * HW + hw;
        float r[4];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float s = ((red[0][lane * E + j] + red[1][lane * E + j]) + red[2][lane * E + j]) + red[3][lane * E + j];
            r[j] = fmaxf(s * a.out_scale, a.clamp_min);
        }
        if constexpr (VEC) store4<T>(op, r);
        else op[0] = from_f<T>(r[0]);
    }
}

// ---- axis 2 : out[b][d][h] = clamp(scale * sum_w f(vol[b][d][h][w])) ----------------------------
// LPR lanes per row (power of two, <= 64); each lane covers 4 contiguous voxels per step when VEC.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void drr_fwd_w_kernel(const DrrArgs a, int lpr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows_per_wave = 64 / lpr;
    const int sub = lane / lpr, li = lane % lpr;
    const int64_t nrows = (int64_t)a.B * a.D * a.H;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave + sub;
    constexpr int E = VEC ? 4 : 1;
    float s = 0.f;
    if (row < nrows) {
        const T* vp = reinterpret_cast<const T*>(a.vol) + row * a.W;
        for (int w = li * E; w < a.W; w += lpr * E) {
            if constexpr (VEC) {
                float v[4];
                load4<T>(vp + w, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) s += drr_f(v[j], a.exp_mode, a.mu);
            } else {
                s += drr_f(to_f<T>(vp[w]), a.exp_mode, a.mu);
            }
        }
    }
    for (int off = lpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (row < nrows && li == 0) {
        const float r = fmaxf(s * a.out_scale, a.clamp_min);
        int64_t oidx = row;
        if (a.transpose_out) {
            const int64_t hh = row % a.H, dd = (row / a.H) % a.D, bb = row / ((int64_t)a.H * a.D);
            oidx = (bb * a.H + hh) * a.D + dd;
        }
        reinterpret_cast<T*>(a.out)[oidx] = from_f<T>(r);
    }
}

// ---- backward (both axes): dvol = dout[proj] * pass(out) * scale * f'(vol) ------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void drr_bwd_kernel(const DrrArgs a) {
    constexpr int E = VEC ? 4 : 1;
    const int64_t total = (int64_t)a.B * a.D * a.H * a.W / E;
    const int64_t HW = (int64_t)a.H * a.W;
    const T* vp = reinterpret_cast<const T*>(a.vol);
    const T* outp = reinterpret_cast<const T*>(a.out);
    const T* dop = reinterpret_cast<const T*>(a.dout);
    T* dvp = reinterpret_cast<T*>(a.dvol);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * E;
        const int64_t w = e % a.W, hh = (e / a.W) % a.H, dd = (e / HW) % a.D, bb = e / (HW * a.D);
        float v[4], g[4];
        if constexpr (VEC) load4<T>(vp + e, v); else v[0] = to_f<T>(vp[e]);
        if (a.axis == 0) {
            const int64_t o = bb * HW + hh * a.W + w;
            float ov[4], dv[4];
            if constexpr (VEC) { load4<T>(outp + o, ov); load4<T>(dop + o, dv); }
            else { ov[0] = to_f<T>(outp[o]); dv[0] = to_f<T>(dop[o]); }
#pragma unroll
            for (int j = 0; j < E; ++j) g[j] = (ov[j] > a.clamp_min) ? dv[j] : 0.f;
        } else {
            const int64_t o = a.transpose_out ? (bb * a.H + hh) * a.D + dd : (bb * a.D + dd) * a.H + hh;
            const float gg = (to_f<T>(outp[o]) > a.clamp_min) ? to_f<T>(dop[o]) : 0.f;
#pragma unroll
            for (int j = 0; j < E; ++j) g[j] = gg;
        }
        float r[4];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const float df = a.exp_mode ? -a.mu * __expf(-a.mu * (v[j] + 1.f)) : 1.f;
            r[j] = g[j] * a.out_scale * df;
        }
        if constexpr (VEC) store4<T>(dvp + e, r); else dvp[e] = from_f<T>(r[0]);
    }
}

template <typename T>
hipError_t fwd(const DrrArgs& a, hipStream_t st) {
    if (a.axis == 0) {
        const int64_t HW = (int64_t)a.H * a.W;
        const bool vec = (HW % 4) == 0;
        const int E = vec ? 4 : 1;
        const int64_t per_b = (HW + 64 * E - 1) / (64 * E);
        dim3 grid((unsigned)(per_b * a.B)), blk(256);
        if (vec) hipLaunchKernelGGL((drr_fwd_d_kernel<T, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((drr_fwd_d_kernel<T, false>), grid, blk, 0, st, a);
    } else if (a.axis == 2) {
        const bool vec = (a.W % 4) == 0;
        int lpr = 1;
        const int want = vec ? a.W / 4 : a.W;
        while (lpr < 64 && lpr < want) lpr <<= 1;
        const int64_t nrows = (int64_t)a.B * a.D * a.H;
        const int rows_per_block = 4 * (64 / lpr);
        dim3 grid((unsigned)((nrows + rows_per_block - 1) / rows_per_block)), blk(256);
        if (vec) hipLaunchKernelGGL((drr_fwd_w_kernel<T, true>), grid, blk, 0, st, a, lpr);
        else hipLaunchKernelGGL((drr_fwd_w_kernel<T, false>), grid, blk, 0, st, a, lpr);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename T>
hipError_t bwd(const DrrArgs& a, hipStream_t st) {
    if (a.axis != 0 && a.axis != 2) return hipErrorInvalidValue;
    const bool vec = (a.W % 4) == 0;
    const int64_t total = (int64_t)a.B * a.D * a.H * a.W / (vec ? 4 : 1);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), blk(256);
    if (vec) hipLaunchKernelGGL((drr_bwd_kernel<T, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((drr_bwd_kernel<T, false>), grid, blk, 0, st, a);
    return hipGetLastError();
}

}  // namespace

hipError_t drr_fwd_launch(const DrrArgs& a, hipStream_t st) { return a.is_bf16 ? fwd<bf16>(a, st) : fwd<float>(a, st); }
hipError_t drr_bwd_launch(const DrrArgs& a, hipStream_t st) { return a.is_bf16 ? bwd<bf16>(a, st) : bwd<float>(a, st); }

}  // namespace hvc
