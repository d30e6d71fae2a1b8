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
// LPR lanes per row (power of two, <= 64); each lane covers 4 contiguous voxels per step when VEC.  A wavefront takes RW groups of
// 64 / LPR rows and requests ALL their loads before the first reduction: at W = 256 a row is exactly one 16-byte load per lane, so
// with one row per wavefront (round 3) a wave had 1 KB in flight and the kernel ran at 0.40 of the HBM roof; RW = 8 keeps 8 KB per
// wave in flight.  The lane-group sums are then folded by wavefront shuffles - no LDS, no atomics.
template <typename T, bool VEC, int RW>
__global__ __launch_bounds__(256) void drr_fwd_w_kernel(const DrrArgs a, int lpr, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows_per_group = 64 / lpr;
    const int sub = lane / lpr, li = lane % lpr;
    const int64_t nrows = (int64_t)a.B * a.D * a.H;
    constexpr int E = VEC ? 4 : 1;
    const T* vbase = reinterpret_cast<const T*>(a.vol);
    // a wavefront owns iters * RW consecutive row groups (a short-lived wave - one batch of loads, one reduction, exit - spends as
    // long being dispatched and setting up addresses as streaming: 8 KB per wave ran at the same 0.40 of the roof as 1 KB)
    for (int it = 0; it < iters; ++it) {
        const int64_t row0 = (((int64_t)blockIdx.x * 4 + wave) * iters + it) * (rows_per_group * RW) + sub;      // row of group u: row0 + u * rows_per_group
        if (row0 - sub >= nrows) break;                  // wave-uniform
        float s[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) s[u] = 0.f;
        for (int w = li * E; w < a.W; w += lpr * E) {
            float v[RW][4];
#pragma unroll
            for (int u = 0; u < RW; ++u) {               // every load of this column step first ...
                const int64_t row = row0 + (int64_t)u * rows_per_group;
                const T* vp = vbase + (row < nrows ? row : nrows - 1) * a.W + w;      // clamped: rows past the end are discarded below
                if constexpr (VEC) load4<T>(vp, v[u]);
                else v[u][0] = to_f<T>(vp[0]);
            }
#pragma unroll
            for (int u = 0; u < RW; ++u)                 // ... then the arithmetic
#pragma unroll
                for (int j = 0; j < E; ++j) s[u] += drr_f(v[u][j], a.exp_mode, a.mu);
        }
#pragma unroll
        for (int u = 0; u < RW; ++u)
            for (int off = lpr >> 1; off > 0; off >>= 1) s[u] += __shfl_xor(s[u], off, 64);
        if (li == 0) {
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int64_t row = row0 + (int64_t)u * rows_per_group;
                if (row >= nrows) continue;
                const float r = fmaxf(s[u] * a.out_scale, a.clamp_min);
                int64_t oidx = row;
                if (a.transpose_out) {
                    const int64_t hh = row % a.H, dd = (row / a.H) % a.D, bb = row / ((int64_t)a.H * a.D);
                    oidx = (bb * a.H + hh) * a.D + dd;
                }
                reinterpret_cast<T*>(a.out)[oidx] = from_f<T>(r);
            }
        }
    }
}

// ---- axis 2, rows of >= 64 vector lanes (W >= 256 elements): tiled form -------------------------------------------------------
// A workgroup takes 4 x 32 rows (wavefront w: 32 rows in four batches of eight, every load of a batch requested before its first
// use), folds a batch's eight lane sums with ONE butterfly (10 shuffles instead of 48: after the xor-32 exchange each half of the
// wave carries four of the eight rows, after xor-16 two, after xor-8 one - the same pairing, level by level, as eight separate
// xor-32 ... xor-1 trees, so the sums are bit-identical) and stores through LDS as 128 consecutive floats.  With the transposed
// output (DRRRenderer's lateral view: out[b][h][d]) the 128 rows are 4 h x 32 d, so that the store is four 128-byte runs along d:
// one 4-byte store per row to addresses 1 KB apart cost 27 % of the kernel (scripts/drr_w_probe.py).
template <typename T>
__global__ __launch_bounds__(256) void drr_fwd_w_tiled_kernel(const DrrArgs a) {
    __shared__ float tile[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* vbase = reinterpret_cast<const T*>(a.vol);
    // row(w, j) for wavefront w, j = 0..31
    int64_t row_w0, row_step, out_base;
    int64_t out_stride_w;                                  // output offset between the tile's four 32-float runs
    if (a.transpose_out) {
        const int nbd = a.D / 32, nbh = a.H / 4;
        const int hb = blockIdx.x % nbh, db = (blockIdx.x / nbh) % nbd, b = blockIdx.x / (nbh * nbd);
        row_w0 = ((int64_t)b * a.D + db * 32) * a.H + hb * 4 + wave;       // (b, d0, h0 + w)
        row_step = a.H;                                                     // next d
        out_base = ((int64_t)b * a.H + hb * 4) * a.D + db * 32;            // out[b][h0 + w][d0 + j]
        out_stride_w = a.D;
    } else {
        row_w0 = (int64_t)blockIdx.x * 128 + wave * 32;
        row_step = 1;
        out_base = (int64_t)blockIdx.x * 128;
        out_stride_w = 32;
    }
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        float s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = 0.f;
        for (int w = lane * 4; w < a.W; w += 256) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) load4<T>(vbase + (row_w0 + (int64_t)(it * 8 + u) * row_step) * a.W + w, v[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[u] += drr_f(v[u][j], a.exp_mode, a.mu);
        }
        // butterfly: 8 -> 4 -> 2 -> 1 values per lane, then three plain levels
        const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
        float t4[4], t2[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float send = h5 ? s[k] : s[k + 4], keep = h5 ? s[k + 4] : s[k];
            t4[k] = keep + __shfl_xor(send, 32, 64);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float send = h4 ? t4[k] : t4[k + 2], keep = h4 ? t4[k + 2] : t4[k];
            t2[k] = keep + __shfl_xor(send, 16, 64);
        }
        float t1 = (h3 ? t2[1] : t2[0]) + __shfl_xor(h3 ? t2[0] : t2[1], 8, 64);
        t1 += __shfl_xor(t1, 4, 64);
        t1 += __shfl_xor(t1, 2, 64);
        t1 += __shfl_xor(t1, 1, 64);
        if ((lane & 7) == 0) tile[wave][it * 8 + ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)] = t1;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int ww = threadIdx.x >> 5, j = threadIdx.x & 31;
        reinterpret_cast<T*>(a.out)[out_base + ww * out_stride_w + j] = from_f<T>(fmaxf(tile[ww][j] * a.out_scale, a.clamp_min));
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
        if (vec && lpr == 64 && (a.W % 256) == 0 && (a.transpose_out ? (a.D % 32 == 0 && a.H % 4 == 0) : nrows % 128 == 0) && nrows >= 128 * 512) {
            hipLaunchKernelGGL((drr_fwd_w_tiled_kernel<T>), dim3((unsigned)(nrows / 128)), dim3(256), 0, st, a);
            return hipGetLastError();
        }
        // eight row groups in flight per wavefront, and as many batches of eight per wavefront as still leave ~8 workgroups per CU
        const int64_t groups = (nrows + (64 / lpr) - 1) / (64 / lpr);
        const int64_t batches = (groups + 7) / 8;                   // wave-batches of RW = 8 row groups
        if (batches >= 4 * 1024) {
            int iters = (int)(batches / (4 * 2048));
            if (iters < 1) iters = 1;
            if (iters > 16) iters = 16;
            const int64_t blocks = (batches + 4 * iters - 1) / (4 * iters);
            dim3 grid((unsigned)blocks), blk(256);
            if (vec) hipLaunchKernelGGL((drr_fwd_w_kernel<T, true, 8>), grid, blk, 0, st, a, lpr, iters);
            else hipLaunchKernelGGL((drr_fwd_w_kernel<T, false, 8>), grid, blk, 0, st, a, lpr, iters);
        } else {
            const int rows_per_block = 4 * (64 / lpr);
            dim3 grid((unsigned)((nrows + rows_per_block - 1) / rows_per_block)), blk(256);
            if (vec) hipLaunchKernelGGL((drr_fwd_w_kernel<T, true, 1>), grid, blk, 0, st, a, lpr, 1);
            else hipLaunchKernelGGL((drr_fwd_w_kernel<T, false, 1>), grid, blk, 0, st, a, lpr, 1);
        }
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
