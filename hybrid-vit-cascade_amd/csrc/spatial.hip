// Spatial data movement for the conv stems and the output head (HBM-bound kernels, gfx950).
//
// Convolutions of the reference
//   models/hybrid_vit_backbone.py:195-210   Conv3d(k3, s1|s2, p1) voxel-embed stem
//   models/diagnostic_losses.py:82-96       Conv2d(k7 s2 p3), Conv2d(k3 s1 p1) X-ray stem
// run as   im2col (this file)  ->  MFMA GEMM with fused bias / pos_embed epilogue (gemm.hip)
// on CHANNELS-LAST activations [B][D][H][W][C] (2-D: D = 1), so the K index (tap, c) of a patch
// row is a run of 16-byte aligned channel vectors, the GEMM output [positions][Cout] is already
// the next layer's channels-last input, and the last stem layer's output IS the (B, N, C) token
// matrix with n = (d*H' + h)*W' + w (reference :255) -- no transpose pass.
// Backward: dcol = dOut W (GEMM, W read in place) -> col2im gather (deterministic, no atomics);
// dW = dOut^T col (split-K GEMM).
//
// Also here: trilinear resize (align_corners=True: reference models/hybrid_vit_backbone.py:272; False: the
// cascade's nn.Upsample / F.interpolate, model_progressive.py:170,211,239,294),
// forward as a gather and backward as an output-stationary gather (each coarse voxel sums the
// fine voxels in its support: deterministic, no atomics).
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

struct Pos { int b, d, h, w; };

// n / d for n < 2^31 by a launch-time constant (hvc_kernels.h FastDiv): the flat-index decodes below otherwise cost two or three 64-bit
// divisions (~100 instructions each) per 16-byte access, more than the interpolation itself
__device__ __forceinline__ uint32_t fdiv32(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.m) + n) >> f.l; }

__device__ __forceinline__ Pos decode(int64_t m, int D, int H, int W) {
    Pos p;
    p.w = (int)(m % W); m /= W;
    p.h = (int)(m % H); m /= H;
    p.d = (int)(m % D); p.b = (int)(m / D);
    return p;
}

// ---- im2col ---------------------------------------------------------------------------------------
// col[m][tap*C + c] = src[b][od*s + kd - p][oh*s + kh - p][ow*s + kw - p][c]  (0 outside), m = output position
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const ConvGeom g, const T* __restrict__ src, T* __restrict__ col) {
    const int taps = g.KD * g.KH * g.KW;
    if constexpr (VEC) {
        // One wavefront per patch row: the output position is decoded once per row (64-bit divisions), the lanes walk the
        // row's 16-byte chunks (tap, 8 channels) with 32-bit arithmetic; stores of a wavefront are one contiguous KiB.
        const int c8n = g.C / 8, nchunk = taps * c8n;
        const int lane = threadIdx.x & 63;
        for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < g.M; m += (int64_t)gridDim.x * 4) {
            const Pos o = decode(m, g.OD, g.OH, g.OW);
            const int64_t sb = (int64_t)o.b * g.SD;
            const int d0 = o.d * g.stride - g.PD, h0 = o.h * g.stride - g.PH, w0 = o.w * g.stride - g.PW;
            T* row = col + m * g.Kp;
            for (int ch = lane; ch < nchunk; ch += 64) {
                const int tap = ch / c8n, c8 = ch - tap * c8n;
                const int kw = tap % g.KW, t2 = tap / g.KW, kh = t2 % g.KH, kd = t2 / g.KH;
                const int sd = d0 + kd, sh = h0 + kh, sw = w0 + kw;
                const bool ok = sd >= 0 && sd < g.SD && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW;
                Chunk8<T> v = zero_chunk<T>();
                if (ok) v = load_chunk<T>(src + (((sb + sd) * g.SH + sh) * g.SW + sw) * g.C + c8 * 8, 8, true);
                T* dst = row + ch * 8;
                if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x8*>(dst) = v.v;
                else { *reinterpret_cast<f32x4*>(dst) = v.a; *reinterpret_cast<f32x4*>(dst + 4) = v.b; }
            }
        }
    } else {
        // Few input channels (the stems' first convolutions, C = 1): one thread builds 8 consecutive k of a patch row,
        // so the output position is decoded once per 16-byte store and the tap arithmetic stays 32-bit.
        const int k8n = (int)(g.Kp / 8), kvalid = taps * g.C;
        const int64_t total = g.M * k8n;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
            const int k0 = (int)(idx % k8n) * 8;
            const int64_t m = idx / k8n;
            const Pos o = decode(m, g.OD, g.OH, g.OW);
            const int64_t sb = (int64_t)o.b * g.SD;
            const int d0 = o.d * g.stride - g.PD, h0 = o.h * g.stride - g.PH, w0 = o.w * g.stride - g.PW;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                v[j] = 0.f;
                if (k < kvalid) {
                    const int c = k % g.C, tap = k / g.C;
                    const int kw = tap % g.KW, kh = (tap / g.KW) % g.KH, kd = tap / (g.KW * g.KH);
                    const int sd = d0 + kd, sh = h0 + kh, sw = w0 + kw;
                    if (sd >= 0 && sd < g.SD && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW)
                        v[j] = to_f<T>(src[(((sb + sd) * g.SH + sh) * g.SW + sw) * g.C + c]);
                }
            }
            T* dst = col + m * g.Kp + k0;
            if constexpr (sizeof(T) == 2) {
                bf16x8 w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = f2bf(v[j]);
                *reinterpret_cast<bf16x8*>(dst) = w;
            } else {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        }
    }
}

// ---- col2im (gather form): dsrc[pos][c] = sum over (m, tap) with src(m, tap) == pos of dcol[m][tap*C + c] ----
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void col2im_kernel(const ConvGeom g, const T* __restrict__ dcol, T* __restrict__ dsrc) {
    constexpr int E = VEC ? 8 : 1;
    const int cn = g.C / E;
    const int64_t total = (int64_t)g.B * g.SD * g.SH * g.SW * cn;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ce = (int)(idx % cn);
        const Pos s = decode(idx / cn, g.SD, g.SH, g.SW);
        float acc[E];
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] = 0.f;
        for (int kd = 0; kd < g.KD; ++kd) {
            const int nd = s.d + g.PD - kd;
            if (nd < 0 || nd % g.stride) continue;
            const int od = nd / g.stride;
            if (od >= g.OD) continue;
            for (int kh = 0; kh < g.KH; ++kh) {
                const int nh = s.h + g.PH - kh;
                if (nh < 0 || nh % g.stride) continue;
                const int oh = nh / g.stride;
                if (oh >= g.OH) continue;
                for (int kw = 0; kw < g.KW; ++kw) {
                    const int nw = s.w + g.PW - kw;
                    if (nw < 0 || nw % g.stride) continue;
                    const int ow = nw / g.stride;
                    if (ow >= g.OW) continue;
                    const int64_t m = (((int64_t)s.b * g.OD + od) * g.OH + oh) * g.OW + ow;
                    const int tap = (kd * g.KH + kh) * g.KW + kw;
                    const T* p = dcol + m * g.Kp + (int64_t)tap * g.C + ce * E;
                    if constexpr (VEC) {
                        Chunk8<T> v = load_chunk<T>(p, 8, true);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += chunk_get<T>(v, j);
                    } else {
                        acc[0] += to_f<T>(p[0]);
                    }
                }
            }
        }
        T* dst = dsrc + (idx / cn) * g.C + ce * E;
        if constexpr (VEC) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
                *reinterpret_cast<bf16x8*>(dst) = o;
            } else {
                *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
            }
        } else {
            dst[0] = from_f<T>(acc[0]);
        }
    }
}

// ---- trilinear resize of single channel volumes [B][d][h][w] -> [B][D][H][W] ------------------------
// Source coordinate of fine index o (ATen area_pixel_compute_source_index):
//   align_corners : o * (in-1)/(out-1)        else : max(0, (o + 0.5) * in/out - 0.5)
struct AxisMap {
    float r;
    int ac;
    __device__ __forceinline__ float src(int o) const { return ac ? r * o : fmaxf(r * (o + 0.5f) - 0.5f, 0.f); }
};
__device__ __forceinline__ AxisMap axis_map(int in, int out, int ac) {
    AxisMap m;
    m.ac = ac;
    m.r = ac ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f) : (float)in / (float)out;
    return m;
}

__global__ __launch_bounds__(256) void trilinear_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int B, int d, int h, int w, int D, int H, int W, int ac) {
    const AxisMap md = axis_map(d, D, ac), mh = axis_map(h, H, ac), mw = axis_map(w, W, ac);
    const int64_t total = (int64_t)B * D * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const Pos o = decode(idx, D, H, W);
        const float fd = md.src(o.d), fh = mh.src(o.h), fw = mw.src(o.w);
        const int d0 = min((int)fd, d - 1), h0 = min((int)fh, h - 1), w0 = min((int)fw, w - 1);
        const int d1 = min(d0 + 1, d - 1), h1 = min(h0 + 1, h - 1), w1 = min(w0 + 1, w - 1);
        const float ld = fd - d0, lh = fh - h0, lw = fw - w0;
        const float* s = src + (int64_t)o.b * d * h * w;
        auto at = [&](int a, int b2, int c) { return s[((int64_t)a * h + b2) * w + c]; };
        const float v = (1 - ld) * ((1 - lh) * ((1 - lw) * at(d0, h0, w0) + lw * at(d0, h0, w1)) + lh * ((1 - lw) * at(d0, h1, w0) + lw * at(d0, h1, w1))) +
                        ld * ((1 - lh) * ((1 - lw) * at(d1, h0, w0) + lw * at(d1, h0, w1)) + lh * ((1 - lw) * at(d1, h1, w0) + lw * at(d1, h1, w1)));
        dst[idx] = v;
    }
}

// W % 4 == 0: one thread = four consecutive outputs of a row.  The index decode, the two outer axis maps and the four source row
// bases are shared by the four outputs, which leave as one 16-byte store (the scalar kernel spends ~60 instructions and a 4-byte
// store per element: instruction-bound at 0.7 TB/s of output on 16 x 256^3).  Same expression per element: bit-identical results.
template <bool FAST>
__global__ __launch_bounds__(256) void trilinear_fwd4_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int B, int d, int h, int w, int D, int H, int W, int ac,
                                                              FastDiv fW4, FastDiv fH, FastDiv fD) {
    const AxisMap md = axis_map(d, D, ac), mh = axis_map(h, H, ac), mw = axis_map(w, W, ac);
    const int W4 = W >> 2;
    const int64_t total = (int64_t)B * D * H * W4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        Pos o;                                    // o.w = group of four along W
        if constexpr (FAST) {                     // total < 2^31
            const uint32_t q1 = fdiv32((uint32_t)idx, fW4), q2 = fdiv32(q1, fH), q3 = fdiv32(q2, fD);
            o.w = (int)((uint32_t)idx - q1 * fW4.d); o.h = (int)(q1 - q2 * fH.d); o.d = (int)(q2 - q3 * fD.d); o.b = (int)q3;
        } else {
            o = decode(idx, D, H, W4);
        }
        const float fd = md.src(o.d), fh = mh.src(o.h);
        const int d0 = min((int)fd, d - 1), h0 = min((int)fh, h - 1);
        const int d1 = min(d0 + 1, d - 1), h1 = min(h0 + 1, h - 1);
        const float ld = fd - d0, lh = fh - h0;
        const float* s = src + (int64_t)o.b * d * h * w;
        const float* r00 = s + ((int64_t)d0 * h + h0) * w;
        const float* r01 = s + ((int64_t)d0 * h + h1) * w;
        const float* r10 = s + ((int64_t)d1 * h + h0) * w;
        const float* r11 = s + ((int64_t)d1 * h + h1) * w;
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float fw = mw.src(4 * o.w + j);
            const int w0 = min((int)fw, w - 1);
            const int w1 = min(w0 + 1, w - 1);
            const float lw = fw - w0;
            out[j] = (1 - ld) * ((1 - lh) * ((1 - lw) * r00[w0] + lw * r00[w1]) + lh * ((1 - lw) * r01[w0] + lw * r01[w1])) +
                     ld * ((1 - lh) * ((1 - lw) * r10[w0] + lw * r10[w1]) + lh * ((1 - lw) * r11[w0] + lw * r11[w1]));
        }
        *reinterpret_cast<f32x4*>(dst + (((int64_t)o.b * D + o.d) * H + o.h) * W + 4 * o.w) = out;
    }
}

// weight of fine index o onto coarse index i along one axis (mirrors the forward's i0 / i1 / lambda)
__device__ __forceinline__ float axis_w(int o, int i, const AxisMap& m, int in) {
    const float f = m.src(o);
    const int i0 = min((int)f, in - 1);
    const int i1 = min(i0 + 1, in - 1);
    const float l = f - i0;
    float wgt = 0.f;
    if (i == i0) wgt += 1.f - l;
    if (i == i1) wgt += l;
    return wgt;
}
// conservative range of fine indices whose support can include coarse index i
__device__ __forceinline__ void axis_range(int i, const AxisMap& m, int out, int& lo, int& hi) {
    if (m.r <= 0.f) { lo = 0; hi = out - 1; return; }
    // src(o) in (i-1, i+1):  align_corners: o in ((i-1)/r, (i+1)/r);  else: o in ((i-0.5)/r - 0.5, (i+1.5)/r - 0.5); +-1 for rounding
    const float a = m.ac ? (i - 1) / m.r : (i - 0.5f) / m.r - 0.5f;
    const float b = m.ac ? (i + 1) / m.r : (i + 1.5f) / m.r - 0.5f;
    lo = max(0, (int)floorf(a) - 1);
    hi = min(out - 1, (int)ceilf(b) + 1);
}

// One wave per coarse voxel: the 64 lanes stride over the fine voxels in its support box and the partial sums
// are folded with a wavefront reduction (fixed order -> deterministic).
__global__ __launch_bounds__(256) void trilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dsrc,
                                                             int B, int d, int h, int w, int D, int H, int W, int ac) {
    const AxisMap md = axis_map(d, D, ac), mh = axis_map(h, H, ac), mw = axis_map(w, W, ac);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = (int64_t)B * d * h * w;
    for (int64_t idx = (int64_t)blockIdx.x * 4 + wave; idx < total; idx += (int64_t)gridDim.x * 4) {
        const Pos s = decode(idx, d, h, w);
        int dlo, dhi, hlo, hhi, wlo, whi;
        axis_range(s.d, md, D, dlo, dhi);
        axis_range(s.h, mh, H, hlo, hhi);
        axis_range(s.w, mw, W, wlo, whi);
        const int nd = dhi - dlo + 1, nh = hhi - hlo + 1, nw = whi - wlo + 1;
        const float* g = dout + (int64_t)s.b * D * H * W;
        float acc = 0.f;
        for (int e = lane; e < nd * nh * nw; e += 64) {
            const int ow = wlo + e % nw, oh = hlo + (e / nw) % nh, od = dlo + e / (nw * nh);
            const float wgt = axis_w(od, s.d, md, d) * axis_w(oh, s.h, mh, h) * axis_w(ow, s.w, mw, w);
            if (wgt != 0.f) acc += wgt * g[((int64_t)od * H + oh) * W + ow];
        }
        acc = wave_sum(acc);
        if (lane == 0) dsrc[idx] = acc;
    }
}

// Adjoint of the 1-D interpolation along one axis of [outer][fine][inner] -> [outer][coarse][inner]; thread per output
// element, inner fastest (coalesced); the ~2/r + 2 fine candidates of a coarse index are walked in a fixed order.
template <bool FAST>
__global__ __launch_bounds__(256) void axis_adjoint_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t outer, int fine,
                                                            int coarse, int64_t inner, int ac, FastDiv fInner, FastDiv fCoarse) {
    const AxisMap m = axis_map(coarse, fine, ac);
    const int64_t total = outer * coarse * inner;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t n, o;
        int i;
        if constexpr (FAST) {                     // total < 2^31
            const uint32_t q1 = fdiv32((uint32_t)idx, fInner), q2 = fdiv32(q1, fCoarse);
            n = (uint32_t)idx - q1 * fInner.d; i = (int)(q1 - q2 * fCoarse.d); o = q2;
        } else {
            n = idx % inner;
            i = (int)((idx / inner) % coarse);
            o = idx / (inner * coarse);
        }
        int lo, hi;
        axis_range(i, m, fine, lo, hi);
        const float* s = src + (o * fine) * inner + n;
        float acc = 0.f;
        for (int f0 = lo; f0 <= hi; f0 += 4) {            // four candidates at a time, loaded without a branch (clamped to the window's end: zero weight)
            float v[4], wg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u;
                v[u] = s[(int64_t)(f <= hi ? f : hi) * inner];
                wg[u] = f <= hi ? axis_w(f, i, m, coarse) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (wg[u] != 0.f) acc += wg[u] * v[u];
        }
        dst[idx] = acc;
    }
}

// The same adjoint along a CONTIGUOUS axis (inner = 1: the W pass, which reads the whole fine gradient - 63 of every 64 bytes of
// the resize backward at a 4x factor), fine % 4 == 0, coarse <= 256, windows of <= 32 fine elements (factors up to ~12): a thread owns one coarse element and takes its candidate
// window as aligned 16-byte loads (3 - 4 per thread; neighbouring threads' windows overlap and hit L1) instead of ~10 dependent
// 4-byte loads at a 16-byte lane stride, and the interpolation weights of the window - the same for every row - come from a table
// the workgroup builds once in LDS (window start + 32 weights per coarse index) instead of ~12 instructions per candidate.  Same
// candidates, same ascending order, same weights (axis_w) as axis_adjoint_kernel: bit-identical sums.  Round 3: 0.14 of the HBM roof.
constexpr int kAdjMaxCoarse = 256, kAdjWin = 32;      // table: 32 KB of LDS
template <bool FAST, int NC>      // NC = 16-byte chunks a window can span (3 / 4 / 6 / 8 for factors <= 2 / 4 / 8 / 12)
__global__ __launch_bounds__(256) void axis_adjoint_w4_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t outer, int fine,
                                                               int coarse, int ac, FastDiv fCoarse) {
    __shared__ __attribute__((aligned(16))) float wtab[kAdjMaxCoarse][kAdjWin];
    __shared__ int c0tab[kAdjMaxCoarse], hitab[kAdjMaxCoarse];
    const AxisMap m = axis_map(coarse, fine, ac);
    for (int i = threadIdx.x; i < coarse; i += 256) {
        int lo, hi;
        axis_range(i, m, fine, lo, hi);
        const int c0 = lo & ~3;
        if (hi > c0 + kAdjWin - 1) hi = c0 + kAdjWin - 1;      // (never on the factors this path serves: the launcher checks)
        c0tab[i] = c0;
        hitab[i] = hi;
#pragma unroll 4
        for (int x = 0; x < kAdjWin; ++x) {
            const int f = c0 + x;
            wtab[i][x] = (f >= lo && f <= hi) ? axis_w(f, i, m, coarse) : 0.f;
        }
    }
    __syncthreads();
    const int64_t total = outer * coarse;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int i;
        int64_t o;
        if constexpr (FAST) {                     // total < 2^31
            const uint32_t q = fdiv32((uint32_t)idx, fCoarse);
            i = (int)((uint32_t)idx - q * fCoarse.d);
            o = q;
        } else {
            i = (int)(idx % coarse);
            o = idx / coarse;
        }
        const int c0 = c0tab[i], hi = hitab[i];
        const float* s = src + o * fine + c0;
        // Every chunk of the window is LOADED without a branch (chunks past the window re-read chunk 0: valid memory, zero weights): with a
        // branch around each load hipcc waits for the loads in flight at every join, and the eight loads of a thread went out one at a time
        // (the W pass ran at 3.2 TB/s, latency-bound).
        f32x4 q[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) q[cc] = *reinterpret_cast<const f32x4*>(s + (c0 + 4 * cc <= hi ? 4 * cc : 0));
        float acc = 0.f;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&wtab[i][4 * cc]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w[j] != 0.f) acc += w[j] * q[cc][j];
        }
        dst[idx] = acc;
    }
}

int grid_for(int64_t work) {
    int64_t blocks = (work + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

hipError_t im2col_launch(const ConvGeom& g, const void* src, void* col, int is_bf16, hipStream_t st) {
    const bool vec = (g.C % 8) == 0;
    const int64_t work = vec ? g.M * 64 : g.M * (g.Kp / 8);      // vec: one wavefront per patch row
    dim3 grid(grid_for(work)), blk(256);
    if (is_bf16) {
        if (vec) hipLaunchKernelGGL((im2col_kernel<bf16, true>), grid, blk, 0, st, g, (const bf16*)src, (bf16*)col);
        else hipLaunchKernelGGL((im2col_kernel<bf16, false>), grid, blk, 0, st, g, (const bf16*)src, (bf16*)col);
    } else {
        if (vec) hipLaunchKernelGGL((im2col_kernel<float, true>), grid, blk, 0, st, g, (const float*)src, (float*)col);
        else hipLaunchKernelGGL((im2col_kernel<float, false>), grid, blk, 0, st, g, (const float*)src, (float*)col);
    }
    return hipGetLastError();
}

hipError_t col2im_launch(const ConvGeom& g, const void* dcol, void* dsrc, int is_bf16, hipStream_t st) {
    const bool vec = (g.C % 8) == 0;
    const int64_t work = (int64_t)g.B * g.SD * g.SH * g.SW * (vec ? g.C / 8 : g.C);
    dim3 grid(grid_for(work)), blk(256);
    if (is_bf16) {
        if (vec) hipLaunchKernelGGL((col2im_kernel<bf16, true>), grid, blk, 0, st, g, (const bf16*)dcol, (bf16*)dsrc);
        else hipLaunchKernelGGL((col2im_kernel<bf16, false>), grid, blk, 0, st, g, (const bf16*)dcol, (bf16*)dsrc);
    } else {
        if (vec) hipLaunchKernelGGL((col2im_kernel<float, true>), grid, blk, 0, st, g, (const float*)dcol, (float*)dsrc);
        else hipLaunchKernelGGL((col2im_kernel<float, false>), grid, blk, 0, st, g, (const float*)dcol, (float*)dsrc);
    }
    return hipGetLastError();
}

hipError_t trilinear_launch(const float* src, float* dst, int B, int d, int h, int w, int D, int H, int W, bool align_corners, bool bwd,
                            hipStream_t st) {
    const int ac = align_corners ? 1 : 0;
    if (!bwd && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int64_t total = (int64_t)B * D * H * (W >> 2);
        const FastDiv fW4 = make_fastdiv((uint32_t)(W >> 2)), fH = make_fastdiv((uint32_t)H), fD = make_fastdiv((uint32_t)D);
        if (total < ((int64_t)1 << 31))
            hipLaunchKernelGGL(trilinear_fwd4_kernel<true>, dim3(grid_for(total)), dim3(256), 0, st, src, dst, B, d, h, w, D, H, W, ac, fW4, fH, fD);
        else
            hipLaunchKernelGGL(trilinear_fwd4_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, src, dst, B, d, h, w, D, H, W, ac, fW4, fH, fD);
    }
    else if (!bwd) hipLaunchKernelGGL(trilinear_fwd_kernel, dim3(grid_for((int64_t)B * D * H * W)), dim3(256), 0, st, src, dst, B, d, h, w, D, H, W, ac);
    else hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(grid_for((int64_t)B * d * h * w * 64)), dim3(256), 0, st, src, dst, B, d, h, w, D, H, W, ac);
    return hipGetLastError();
}

int64_t trilinear_bwd_workspace_floats(int B, int d, int h, int w, int D, int H, int W) {
    return (int64_t)B * D * H * w + (int64_t)B * D * h * w;      // after the W pass | after the H pass
}

hipError_t trilinear_bwd_separable_launch(const float* dout, float* dsrc, float* workspace, int B, int d, int h, int w, int D, int H, int W,
                                          bool align_corners, hipStream_t st) {
    const int ac = align_corners ? 1 : 0;
    float* t1 = workspace;                                  // [B*D*H][w]
    float* t2 = workspace + (int64_t)B * D * H * w;         // [B*D][h][w]
    // window of a coarse index: at most 2 W / w + 4 fine candidates (axis_range) plus 3 of alignment
    auto adjoint = [&](const float* in, float* out, int64_t outer, int fine, int coarse, int64_t inner) {
        const int64_t total = outer * coarse * inner;
        const FastDiv fi = make_fastdiv((uint32_t)inner), fc = make_fastdiv((uint32_t)coarse);
        if (total < ((int64_t)1 << 31) && inner < ((int64_t)1 << 31))
            hipLaunchKernelGGL(axis_adjoint_kernel<true>, dim3(grid_for(total)), dim3(256), 0, st, in, out, outer, fine, coarse, inner, ac, fi, fc);
        else
            hipLaunchKernelGGL(axis_adjoint_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, in, out, outer, fine, coarse, inner, ac, fi, fc);
    };
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15u) == 0 && w <= kAdjMaxCoarse && 2 * ((W + w - 1) / w) + 7 <= kAdjWin) {
        const int64_t total = (int64_t)B * D * H * w;
        const FastDiv fc = make_fastdiv((uint32_t)w);
        const int span = 2 * ((W + w - 1) / w) + 7;           // fine candidates of a coarse index + alignment (the launch condition above: <= 32)
        const int nc = (span + 3) / 4;
        const dim3 grid(grid_for(total));
#define HVC_ADJ_W4(FASTV, NCV) hipLaunchKernelGGL((axis_adjoint_w4_kernel<FASTV, NCV>), grid, dim3(256), 0, st, dout, t1, (int64_t)B * D * H, W, w, ac, fc)
        if (total < ((int64_t)1 << 31)) {
            if (nc <= 3) HVC_ADJ_W4(true, 3); else if (nc <= 4) HVC_ADJ_W4(true, 4); else if (nc <= 6) HVC_ADJ_W4(true, 6); else HVC_ADJ_W4(true, 8);
        } else {
            HVC_ADJ_W4(false, 8);
        }
#undef HVC_ADJ_W4
    } else {
        adjoint(dout, t1, (int64_t)B * D * H, W, w, 1);
    }
    adjoint(t1, t2, (int64_t)B * D, H, h, w);
    adjoint(t2, dsrc, (int64_t)B, D, d, (int64_t)h * w);
    return hipGetLastError();
}

}  // namespace hvc
