// Single-channel convolutions of the cascade glue, written as streaming kernels (HBM-bound: one side of each is a 1-channel volume,
// the other a 32- / 64-channel one):
//   direct_regression/progressive_cascade/model_progressive.py:171,240,260   Conv3d(1 -> 32 | 64, k3, p1)      upsample_from_* / detail_enhancer[0]
//   models/hybrid_vit_backbone.py:199                                         Conv3d(1 -> C/4, k3, s2, p1)      first voxel-embed layer of the direct model
//   direct_regression/progressive_cascade/model_progressive.py:266           Conv3d(32 -> 1, k1)               detail_enhancer[-1]
// Through im2col + GEMM (spatial.hip, gemm.hip) the Cin = 1 layers wrote and re-read a [voxels][32] patch matrix as large as their own
// output (1 GiB at 256^3) and kept it for the backward; the Cout = 1 layer ran a 128-wide MFMA tile for one output column and its input
// gradient went through a [voxels][32] dcol matrix and col2im.  Here:
//   * conv_c1_fwd : a workgroup stages the halo of a 4 x 8 x 32 output block of the 1-channel volume in LDS (4 KB); each wavefront takes rows
//                   of 32 output voxels, gathers their 27 taps (padded to 32) from the halo into an MFMA B operand and multiplies by the
//                   [Cout][32] weight fragment held in registers (two 32x32x16 MFMAs per 32 output channels); the tile goes through a
//                   wavefront-private LDS stage so that the channels-last stores are 16 bytes per lane, 1 - 2 KiB contiguous per row.
//   * conv_c1_dw  : the same halo and tap gather, now as the B operand of dW[co][tap] = sum_voxels dy[voxel][co] patch[voxel][tap]; dy rows
//                   are streamed once (the only HBM traffic that matters), transposed through LDS (ds_read_b64_tr_b16).  Tap 27 of the
//                   padded patch is the constant 1, so column 27 of the result is the bias gradient.  Per-workgroup fp32 partials, summed
//                   in a fixed order by a second kernel (deterministic, as the split-K weight gradients of gemm.hip).
//   * conv_c1_dx  : the dcol tile of the input gradient stays in LDS (MFMA per halo voxel, then a 27-entry gather per output voxel).
//   * conv_1x1_o1_fwd / _bwd : row dot products / outer products, 16 bytes per lane.
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

constexpr int kTZ = 4, kTY = 8, kTX = 32;     // output voxels per workgroup: 32 rows (4 x 8) of 32
constexpr int kTaps = 27, kTapPad = 32;

template <int STRIDE> struct Halo {
    static constexpr int HZ = STRIDE * (kTZ - 1) + 3, HY = STRIDE * (kTY - 1) + 3, HX = STRIDE * (kTX - 1) + 3;
    static constexpr int HXP = (HX + 1) & ~1;
    static constexpr int N = HZ * HY * HXP;
    static constexpr int tap_off(int t) { return ((t / 9) * HY + (t / 3) % 3) * HXP + t % 3; }      // t = kd * 9 + kh * 3 + kw
};

struct TileId { int b, z0, y0, x0; };

__device__ __forceinline__ TileId tile_of(const ConvC1Args& a, int t) {
    TileId id;
    id.x0 = (t % a.tiles_x) * kTX; t /= a.tiles_x;
    id.y0 = (t % a.tiles_y) * kTY; t /= a.tiles_y;
    id.z0 = (t % a.tiles_z) * kTZ;
    id.b = t / a.tiles_z;
    return id;
}

// the input halo of one output block -> LDS (zeros outside the volume: the layer's zero padding)
template <int STRIDE>
__device__ __forceinline__ void load_halo(const ConvC1Args& a, const TileId& id, bf16* halo, int tid) {
    using H = Halo<STRIDE>;
    const bf16* xb = reinterpret_cast<const bf16*>(a.x) + (int64_t)id.b * a.SD * a.SH * a.SW;
    const int sz0 = id.z0 * STRIDE - 1, sy0 = id.y0 * STRIDE - 1, sx0 = id.x0 * STRIDE - 1;
    for (int e = tid; e < H::N; e += 256) {
        const int hx = e % H::HXP, r = e / H::HXP, hy = r % H::HY, hz = r / H::HY;
        const int sz = sz0 + hz, sy = sy0 + hy, sx = sx0 + hx;
        const bool ok = sz >= 0 && sz < a.SD && sy >= 0 && sy < a.SH && sx >= 0 && sx < a.SW;
        const bf16 v = xb[ok ? ((int64_t)sz * a.SH + sy) * a.SW + sx : 0];
        halo[e] = ok ? v : (bf16)0.f;
    }
}

__device__ __forceinline__ uint32_t pack2(bf16 lo, bf16 hi) {
    return (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
}
__device__ __forceinline__ bf16x8 from_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(bf16x8, (u32x4){w0, w1, w2, w3});
}

// ---- forward ----------------------------------------------------------------------------------------------------------------
template <int STRIDE, int NT>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const ConvC1Args a) {
    using H = Halo<STRIDE>;
    constexpr int ROWB = 64 * NT + 16;                      // bytes per voxel row of the store stage (16-byte aligned, off the bank period)
    __shared__ __attribute__((aligned(16))) bf16 halo[H::N];
    __shared__ __attribute__((aligned(16))) char stage_all[4 * kTX * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const TileId id = tile_of(a, blockIdx.x);
    load_halo<STRIDE>(a, id, halo, tid);

    // weight fragments (A operand: lane <-> output channel, k <-> tap) and the bias of this lane's accumulator rows
    const bf16* w = reinterpret_cast<const bf16*>(a.w2d);
    bf16x8 wf[NT][2];
    float bs[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) wf[nt][s] = *reinterpret_cast<const bf16x8*>(w + (size_t)(32 * nt + r) * kTapPad + 16 * s + 8 * h);
        if (h) {                    // taps 27..31 do not exist: elements 3..7 of the upper half of k-step 1
#pragma unroll
            for (int j = 3; j < 8; ++j) wf[nt][1][j] = (bf16)0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) bs[nt][i] = a.bias ? a.bias[32 * nt + acc_row(i, h)] : 0.f;
    }
    // per-lane halo offsets of this lane's 16 taps (k-step s, element j <-> tap 16 s + 8 h + j), in elements, relative to the row's origin
    int toff[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t0 = 16 * s + j, t1 = 16 * s + 8 + j;
            const int o1 = t1 < kTaps ? H::tap_off(t1) : 0;
            toff[s][j] = (h ? o1 : H::tap_off(t0)) + r * STRIDE;
        }
    const uint32_t m23 = h ? 0x0000ffffu : 0xffffffffu, m47 = h ? 0u : 0xffffffffu;
    char* stage = stage_all + wave * (kTX * ROWB);
    __syncthreads();

    bf16* yb = reinterpret_cast<bf16*>(a.y);
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i, z = row / kTY, yy = row % kTY;
        const int oz = id.z0 + z, oy = id.y0 + yy;
        if (oz >= a.OD || oy >= a.OH) continue;                                  // wave-uniform
        const bf16* hp = halo + ((z * STRIDE) * H::HY + yy * STRIDE) * H::HXP;
        uint32_t pw[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) pw[s][j >> 1] = pack2(hp[toff[s][j]], hp[toff[s][j + 1]]);
        pw[1][1] &= m23; pw[1][2] &= m47; pw[1][3] &= m47;
        const bf16x8 p0 = from_words(pw[0][0], pw[0][1], pw[0][2], pw[0][3]), p1 = from_words(pw[1][0], pw[1][1], pw[1][2], pw[1][3]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = bs[nt][e];
            acc = mfma32(wf[nt][0], p0, acc);
            acc = mfma32(wf[nt][1], p1, acc);
            // rows = output channel (registers), column = voxel (lane): four consecutive channels per register group
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t lo = pack2(f2bf(acc[4 * g]), f2bf(acc[4 * g + 1])), hi = pack2(f2bf(acc[4 * g + 2]), f2bf(acc[4 * g + 3]));
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(stage + r * ROWB + (32 * nt + 8 * g + 4 * h) * 2) = (u32x2){lo, hi};
            }
        }
        // 32 voxels x (64 NT) bytes, contiguous in the channels-last output: 16 bytes per lane
        bf16* yrow = yb + ((((int64_t)id.b * a.OD + oz) * a.OH + oy) * a.OW + id.x0) * a.Cout;
#pragma unroll
        for (int k = 0; k < 2 * NT; ++k) {
            const int c = lane + 64 * k, vox = c / (4 * NT), part = c % (4 * NT);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + vox * ROWB + part * 16);
            if (id.x0 + vox < a.OW) *reinterpret_cast<bf16x8*>(yrow + vox * a.Cout + part * 8) = v;
        }
    }
}

// ---- weight (and bias) gradient ------------------------------------------------------------------------------------------------
template <int STRIDE, int NT>
__global__ __launch_bounds__(256) void conv_c1_dw_kernel(const ConvC1Args a) {
    using H = Halo<STRIDE>;
    constexpr int CW = 32 * NT;                              // dy tile width (output channels)
    __shared__ __attribute__((aligned(16))) bf16 halo[H::N];
    __shared__ __attribute__((aligned(16))) bf16 dyt_all[4 * kTX * CW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    bf16* dyt = dyt_all + wave * (kTX * CW);
    // B operand: lane <-> tap r, k <-> voxel 16 s + 8 h + j of the row
    const bool one = r == kTaps;                              // tap 27: the constant 1 (its dW column is the bias gradient)
    const int tbase = (r < kTaps ? H::tap_off(r) : 0) + 8 * h * STRIDE;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    const bf16* dyb = reinterpret_cast<const bf16*>(a.dy);
    const uint32_t ones = 0x3f803f80u;

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const TileId id = tile_of(a, t);
        __syncthreads();                                      // the previous tile's halo has been read
        load_halo<STRIDE>(a, id, halo, tid);
        __syncthreads();
        // dy rows of the wavefront's 8 output rows, one row ahead in registers (two rows in flight per wavefront)
        auto load_row = [&](int i, bf16x8 (&v)[2 * NT]) {
            const int row = wave * 8 + i, oz = id.z0 + row / kTY, oy = id.y0 + row % kTY;
            const bool in = i < 8 && oz < a.OD && oy < a.OH;
            const bf16* drow = dyb + ((((int64_t)id.b * a.OD + (in ? oz : 0)) * a.OH + (in ? oy : 0)) * a.OW + id.x0) * a.Cout;
#pragma unroll
            for (int k = 0; k < 2 * NT; ++k) {
                const int c = lane + 64 * k, vox = c / (4 * NT), part = c % (4 * NT);
                const bool ok = in && id.x0 + vox < a.OW;                    // zeros past the end of the row; the load itself carries no branch
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(drow + (ok ? vox * a.Cout + part * 8 : 0));
                v[k] = ok ? t : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        };
        bf16x8 cur[2 * NT], nxt[2 * NT];
        load_row(0, cur);
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            load_row(i + 1, nxt);
            const int row = wave * 8 + i, z = row / kTY, yy = row % kTY;
            // -> LDS tile [voxel][co] (rows outside the volume are zeros: they add nothing)
#pragma unroll
            for (int k = 0; k < 2 * NT; ++k) {
                const int c = lane + 64 * k;
                tile_store<CW>(dyt, c / (4 * NT), c % (4 * NT), cur[k]);
            }
            const bf16* hp = halo + ((z * STRIDE) * H::HY + yy * STRIDE) * H::HXP + tbase;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint32_t pw[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) pw[j >> 1] = one ? ones : pack2(hp[(16 * s + j) * STRIDE], hp[(16 * s + j + 1) * STRIDE]);
                const bf16x8 pf = from_words(pw[0], pw[1], pw[2], pw[3]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 df = tr_frag<CW, false>(dyt, 16 * s, 32 * nt, lane);      // lane <-> co, k <-> voxel
                    acc[nt] = mfma32(df, pf, acc[nt]);
                }
            }
#pragma unroll
            for (int k = 0; k < 2 * NT; ++k) cur[k] = nxt[k];
        }
    }
    // rows = co (registers), column = tap (lane).  The four wavefronts' tiles are added in wavefront order through LDS (the dy tiles
    // are free now), so the second pass sums one partial per workgroup.
    float* red = reinterpret_cast<float*>(dyt_all);
    static_assert(sizeof(float) * 32 * NT * kTapPad <= sizeof(bf16) * 4 * kTX * CW, "the partial tile fits the dy tiles");
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* p = red + (32 * nt + acc_row(e, h)) * kTapPad + r;
                    const float v = w == 0 ? acc[nt][e] : *p + acc[nt][e];
                    if (w == 3) a.workspace[(size_t)blockIdx.x * (a.Cout * kTapPad) + (32 * nt + acc_row(e, h)) * kTapPad + r] = v;
                    else *p = v;
                }
        }
    }
}

// ---- input gradient of the Cin = 1 layers (stride 1):  dx[p] = sum_tap sum_co dy[p - (tap - 1)][co] w[co][tap] ---------------------------
// Through hvc_gemm + hvc_col2im this was a [voxels][32] dcol matrix written and gathered again (0.67 + 2.17 ms at 256^3 x 32).  Here the dcol
// tile stays in LDS: a workgroup takes a 2 x 8 x 32 block of dx, computes T[q][tap] = dy[q] . w[:, tap] for the 4 x 10 x 34 voxels q of the block's
// halo with MFMAs (A = w^T: lane <-> tap, B = dy rows straight from global memory: lane <-> voxel; bf16 results, as the dcol matrix was), and
// every thread then sums the 27 entries T[p - (tap - 1)][tap] of its own output voxel.
constexpr int kDZ = 2, kDY = 8, kDX = 32;                      // 512 outputs per workgroup (one per thread), halo read 2.66 x
constexpr int kDHZ = kDZ + 2, kDHY = kDY + 2, kDHX = kDX + 2, kDHN = kDHZ * kDHY * kDHX;      // 1360 halo voxels
constexpr int kTRow = 28;                                      // bf16 per T row: taps 0..27 (56-byte stride: 14 banks, conflict-free 8-byte stores)

template <int NT>
__global__ __launch_bounds__(512, 2) void conv_c1_dx_kernel(const ConvC1Args a) {
    constexpr int CO = 32 * NT, KS = CO / 16, NTILE = (kDHN + 31) / 32;
    __shared__ __attribute__((aligned(16))) bf16 T[NTILE * 32 * kTRow];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    int t = blockIdx.x;
    const int x0 = (t % a.tiles_x) * kDX; t /= a.tiles_x;
    const int y0 = (t % a.tiles_y) * kDY; t /= a.tiles_y;
    const int z0 = (t % a.tiles_z) * kDZ;
    const int b = t / a.tiles_z;
    const bf16* dyb = reinterpret_cast<const bf16*>(a.dy) + (int64_t)b * a.OD * a.OH * a.OW * CO;
    // w^T fragments: lane <-> tap r (rows 27.. of the [32][CO] operand are zero), k <-> output channel
    const bf16* wt = reinterpret_cast<const bf16*>(a.w2d);
    bf16x8 wf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(wt + r * CO + 16 * s + 8 * h);

    auto fetch = [&](int tile, bf16x8 (&v)[KS]) {               // dy row of halo voxel 32 tile + r, this lane's halves
        const int hv = 32 * tile + r;
        const int hx = hv % kDHX, q = hv / kDHX, hy = q % kDHY, hz = q / kDHY;
        const int sz = z0 - 1 + hz, sy = y0 - 1 + hy, sx = x0 - 1 + hx;
        const bool ok = tile < NTILE && hv < kDHN && sz >= 0 && sz < a.OD && sy >= 0 && sy < a.OH && sx >= 0 && sx < a.OW;
        const bf16* row = dyb + (((int64_t)(ok ? sz : 0) * a.OH + (ok ? sy : 0)) * a.OW + (ok ? sx : 0)) * CO + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(row + 16 * s);      // (row is a valid address either way: no branch around the load)
            v[s] = ok ? t : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    bf16x8 cur[KS], nxt[KS];
    fetch(wave, cur);
#pragma unroll 1
    for (int tile = wave; tile < NTILE; tile += 8) {
        fetch(tile + 8, nxt);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma32(wf[s], cur[s], acc);
        // rows = tap (registers), column = voxel (lane): four consecutive taps per register group -> T[voxel][tap] (taps 28..31 do not exist)
        bf16* trow = T + (32 * tile + r) * kTRow;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if (g < 3 || h == 0)
                *reinterpret_cast<u32x2*>(trow + 8 * g + 4 * h) =
                    (u32x2){pack2(f2bf(acc[4 * g]), f2bf(acc[4 * g + 1])), pack2(f2bf(acc[4 * g + 2]), f2bf(acc[4 * g + 3]))};
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
    }
    __syncthreads();
    const int x = tid & 31, y = (tid >> 5) & 7, z = tid >> 8;
    float sum = 0.f;
#pragma unroll
    for (int tap = 0; tap < kTaps; ++tap) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        sum += bf2f(T[(((z + 2 - kd) * kDHY + (y + 2 - kh)) * kDHX + (x + 2 - kw)) * kTRow + tap]);
    }
    const int oz = z0 + z, oy = y0 + y, ox = x0 + x;
    if (oz < a.OD && oy < a.OH && ox < a.OW)
        reinterpret_cast<bf16*>(a.y)[(((int64_t)b * a.OD + oz) * a.OH + oy) * a.OW + ox] = f2bf(sum);
}

// dw[e] = sum over the partials, fixed order: a block takes 16 consecutive outputs, its 256 threads = 16 outputs x 16 row groups
__global__ __launch_bounds__(256) void conv_c1_dw_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int n, int parts) {
    __shared__ float red[16][16];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (e < n) {
        int p = rg;
        for (; p + 16 < parts; p += 32) { s0 += ws[(size_t)p * n + e]; s1 += ws[(size_t)(p + 16) * n + e]; }
        if (p < parts) s0 += ws[(size_t)p * n + e];
    }
    red[rg][cl] = s0 + s1;
    __syncthreads();
    if (rg == 0 && e < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        dw[e] = s;
    }
}

// ---- 1 x 1 x 1, one output channel ---------------------------------------------------------------------------------------------------
// y[m] = bias + sum_c x[m][c] w[c]: CH = C / 8 lanes per row, 16 bytes each
template <int CH>
__global__ __launch_bounds__(256) void conv_o1_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const float* __restrict__ bias,
                                                          bf16* __restrict__ y, int64_t M) {
    const int part = threadIdx.x % CH;
    float wv[8];
    {
        const bf16x8 wc = *reinterpret_cast<const bf16x8*>(w + part * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = bf2f(wc[j]);
    }
    const float b0 = bias ? bias[0] : 0.f;
    const int64_t total = M * CH, step = (int64_t)gridDim.x * 256;
    constexpr int U = 4;
    for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < total; base += step * U) {
        bf16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = base + u * step;
            v[u] = c < total ? *reinterpret_cast<const bf16x8*>(x + c * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s = __builtin_fmaf(bf2f(v[u][j]), wv[j], s);
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) s += __shfl_xor(s, o);
            const int64_t c = base + u * step;
            if (part == 0 && c < total) y[c / CH] = f2bf(s + b0);
        }
    }
}

// dx[m][c] = dy[m] w[c];  partial dw[c] = sum_m dy[m] x[m][c], db = sum_m dy[m] per workgroup (fixed-order second pass)
template <int CH>
__global__ __launch_bounds__(256) void conv_o1_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, const bf16* __restrict__ w,
                                                          bf16* __restrict__ dx, float* __restrict__ ws, int64_t M) {
    __shared__ float red[256][9];
    const int part = threadIdx.x % CH;
    float wv[8], dw[8], db = 0.f;
    {
        const bf16x8 wc = *reinterpret_cast<const bf16x8*>(w + part * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { wv[j] = bf2f(wc[j]); dw[j] = 0.f; }
    }
    const int64_t total = M * CH, step = (int64_t)gridDim.x * 256;
    constexpr int U = 4;
    for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < total; base += step * U) {
        bf16x8 v[U];
        float g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = base + u * step;
            const bool ok = c < total;
            v[u] = ok ? *reinterpret_cast<const bf16x8*>(x + c * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            g[u] = ok ? bf2f(dy[c / CH]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = base + u * step;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dw[j] = __builtin_fmaf(g[u], bf2f(v[u][j]), dw[j]);
                o[j] = f2bf(g[u] * wv[j]);
            }
            if (part == 0) db += g[u];
            if (dx && c < total) *reinterpret_cast<bf16x8*>(dx + c * 8) = o;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = dw[j];
    red[threadIdx.x][8] = db;
    __syncthreads();
    // threads 0 .. 8 CH: one output each, summed over the 256 / CH threads of its part in a fixed order
    if (threadIdx.x < 8 * CH + 1) {
        const int o = threadIdx.x;
        const int p = o < 8 * CH ? o / 8 : 0, j = o < 8 * CH ? o % 8 : 8;
        float s = 0.f;
        for (int t = p; t < 256; t += CH) s += red[t][j];
        ws[(size_t)blockIdx.x * (8 * CH + 1) + o] = s;
    }
}

// ---- 3 x 3 x 3, stride 1, 32 | 64 -> 32 | 64 channels: LDS halo tile ------------------------------------------------------------------
// (model_progressive.py:263 Conv3d(64, 32, 3, padding=1) at 256^3 and its input gradient.)  As an implicit GEMM every tap re-fetches its
// 128-byte channel vectors from L2: 27 x the input, 56 GB per pass at 256^3, and the pass runs at L2 speed (3.8 ms).  Here a workgroup
// stages the halo of a 2 x 8 x 32 output block once per 16-channel slice (4 x 10 x 34 voxels x 32 B, slots 48 B apart: the 16 slots a
// ds_read_b128 group touches land on distinct banks) and every tap reads its patch fragment from LDS at a constant offset.
// With 32 output channels a patch fragment feeds ONE MFMA per tap - 1 KiB of LDS per 32-cycle MFMA is all the LDS delivers - so the
// fragment is reused across the kh taps instead: a wavefront owns four consecutive output rows (same z) and a halo row read once for
// (kd, kw) goes into up to three of them (kh = 0, 1, 2) against three register-resident weight fragments: 6 reads per 12 MFMAs.
// Weight fragments come from global memory pre-arranged lane-linear by the host (1 KiB per load).  Accumulator rows = output channel,
// column = voxel; stores go through LDS as in conv_c1_fwd.
constexpr int kBZ = 2, kBY = 8, kBX = 32;                     // output block
constexpr int kHZ = kBZ + 2, kHY = kBY + 2, kHX = kBX + 2;    // its halo: 4 x 10 x 34 = 1360 voxels
constexpr int kHSlot = 24;                                    // bf16 per halo voxel slot: 16 channels + 8 (48-byte stride)

template <int CI, int CO>
__global__ __launch_bounds__(256, 2) void conv3_halo_kernel(const Conv3Args a) {
    constexpr int NCK = CI / 16, NT = CO / 32;
    constexpr int ROWB = 64 * NT + 16;                        // store stage: bytes per voxel
    static_assert(4 * kBX * ROWB <= kHZ * kHY * kHX * kHSlot * 2, "the store stage fits the halo");
    __shared__ __attribute__((aligned(16))) bf16 halo[kHZ * kHY * kHX * kHSlot];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    int t = blockIdx.x;
    const int x0 = (t % a.tiles_x) * kBX; t /= a.tiles_x;
    const int y0 = (t % a.tiles_y) * kBY; t /= a.tiles_y;
    const int z0 = (t % a.tiles_z) * kBZ;
    const int b = t / a.tiles_z;
    const bf16* xb = reinterpret_cast<const bf16*>(a.x) + (int64_t)b * a.D * a.H * a.W * CI;
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(a.wfrag) + lane;
    const int zw = wave >> 1, yw = 4 * (wave & 1);            // this wavefront's rows: (zw, yw + j), j = 0..3

    f32x16 acc[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][nt][e] = a.bias ? a.bias[32 * nt + acc_row(e, h)] : 0.f;
    // this lane's patch origin: halo voxel (zw, yw, r) + its 8-channel half
    const bf16* pbase = halo + ((zw * kHY + yw) * kHX + r) * kHSlot + 8 * h;

    // weight fragments [tap][ck][nt][lane]: the three kh taps of one (kd, kw), one group ahead in registers
    auto load_w = [&](int ck, int g, bf16x8 (&w)[3][NT]) {      // g = kd * 3 + kw
        const int kd = g / 3, kw = g % 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w[kh][nt] = wf[(size_t)((((kd * 3 + kh) * 3 + kw) * NCK + ck) * NT + nt) * 64];
    };

    // The slice's weight fragments sit PD groups ahead in registers (all nine at 32 output channels): a group's loads have PD x 12
    // MFMAs - or, for the first ones, the halo staging and its barrier - to arrive; one group ahead they did not (L2 latency > 12 MFMAs).
    // At 32 output channels the registers also hold the NEXT slice's halo pieces, requested right after the barrier that publishes the
    // current slice, so that their latency runs under the slice's 108 MFMAs (AHEAD); at 64 the accumulators leave no room and a slice's
    // pieces are fetched in two batches between the barriers.
    constexpr bool AHEAD = NT == 1;
    constexpr int PD = NT == 1 ? 3 : 2;
    constexpr int HN = kHZ * kHY * kHX * 2;                     // 16-byte halo pieces per slice
    constexpr int HITER = (HN + 255) / 256, HB = AHEAD ? HITER : (HITER + 1) / 2;
    bf16x8 hreg[HB];
    auto halo_fetch = [&](int ck, int first) {                  // pieces first .. first + HB - 1 of this thread -> registers
#pragma unroll
        for (int it = 0; it < HB; ++it) {
            const int e = tid + 256 * (first + it);
            const int piece = e & 1, hv = e >> 1;
            const int hx = hv % kHX, q = hv / kHX, hy = q % kHY, hz = q / kHY;
            const int sz = z0 - 1 + hz, sy = y0 - 1 + hy, sx = x0 - 1 + hx;
            // loaded without a branch (out-of-volume pieces read the block's first voxel and are zeroed by a select): hipcc serialises loads
            // that sit behind divergent branches
            const bool ok = e < HN && sz >= 0 && sz < a.D && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
            const int64_t off = ok ? (((int64_t)sz * a.H + sy) * a.W + sx) * CI + 16 * ck + 8 * piece : 0;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + off);
            hreg[it] = ok ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto halo_commit = [&](int first) {
#pragma unroll
        for (int it = 0; it < HB; ++it) {
            const int e = tid + 256 * (first + it);
            if (e < HN) *reinterpret_cast<bf16x8*>(halo + (e >> 1) * kHSlot + 8 * (e & 1)) = hreg[it];
        }
    };
    if constexpr (AHEAD) halo_fetch(0, 0);
#pragma unroll 1
    for (int ck = 0; ck < NCK; ++ck) {
        bf16x8 wbuf[PD][3][NT];
#pragma unroll
        for (int g = 0; g < PD; ++g) load_w(ck, g, wbuf[g]);
        if constexpr (AHEAD) {
            if (ck) __syncthreads();                           // the previous slice has been read
            halo_commit(0);
        } else {
            halo_fetch(ck, 0);
            if (ck) __syncthreads();
            halo_commit(0);
            halo_fetch(ck, HB);
            halo_commit(HB);
        }
        __syncthreads();
        if constexpr (AHEAD) {
            if (ck + 1 < NCK) halo_fetch(ck + 1, 0);
        }
        // patch fragments of group g + 1 are requested ahead of group g's MFMAs (AHEAD; hipcc otherwise issues each read directly in
        // front of its first MFMA and the wavefront sits through the LDS latency six times per group)
        auto read_patches = [&](int g, bf16x8 (&pf)[6]) {
            const bf16* pg = pbase + ((g / 3) * kHY * kHX + g % 3) * kHSlot;
#pragma unroll
            for (int i = 0; i < 6; ++i) pf[i] = *reinterpret_cast<const bf16x8*>(pg + i * (kHX * kHSlot));
        };
        bf16x8 pfc[6], pfn[6];
        read_patches(0, pfc);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if constexpr (AHEAD) {
                if (g + 1 < 9) read_patches(g + 1, pfn);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int j = i - kh;
                    if (j >= 0 && j < 4) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[j][nt] = mfma32(wbuf[g % PD][kh][nt], pfc[i], acc[j][nt]);
                    }
                }
            if (g + PD < 9) load_w(ck, g + PD, wbuf[g % PD]);
            if (g + 1 < 9) {
                if constexpr (AHEAD) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) pfc[i] = pfn[i];
                } else {
                    read_patches(g + 1, pfc);
                }
            }
        }
    }
    __syncthreads();                                            // the halo is free: it becomes the store stage
    char* st = reinterpret_cast<char*>(halo) + wave * (kBX * ROWB);
    bf16* yb = reinterpret_cast<bf16*>(a.y);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oz = z0 + zw, oy = y0 + yw + j;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t lo = pack2(f2bf(acc[j][nt][4 * g]), f2bf(acc[j][nt][4 * g + 1])), hi = pack2(f2bf(acc[j][nt][4 * g + 2]), f2bf(acc[j][nt][4 * g + 3]));
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(st + r * ROWB + (32 * nt + 8 * g + 4 * h) * 2) = (u32x2){lo, hi};
            }
        if (oz >= a.D || oy >= a.H) continue;                   // wave-uniform
        bf16* yrow = yb + ((((int64_t)b * a.D + oz) * a.H + oy) * a.W + x0) * CO;
#pragma unroll
        for (int k = 0; k < 2 * NT; ++k) {
            const int cc = lane + 64 * k, vox = cc / (4 * NT), part = cc % (4 * NT);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(st + vox * ROWB + part * 16);
            if (x0 + vox < a.W) *reinterpret_cast<bf16x8*>(yrow + vox * CO + part * 8) = v;
        }
    }
}

template <typename K>
void launch_c1(K kernel, int grid, const ConvC1Args& a, hipStream_t st) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, st, a); }

}  // namespace

bool conv_c1_supported(int Cout, int stride) { return (Cout == 32 || Cout == 64) && (stride == 1 || stride == 2); }

static void fill_tiles(ConvC1Args& a) {
    a.OD = (a.SD - 1) / a.stride + 1; a.OH = (a.SH - 1) / a.stride + 1; a.OW = (a.SW - 1) / a.stride + 1;      // k3 p1
    a.tiles_x = (a.OW + kTX - 1) / kTX; a.tiles_y = (a.OH + kTY - 1) / kTY; a.tiles_z = (a.OD + kTZ - 1) / kTZ;
    a.ntiles = a.B * a.tiles_z * a.tiles_y * a.tiles_x;
}

int conv_c1_dw_parts(int B, int SD, int SH, int SW, int stride) {
    ConvC1Args a{};
    a.B = B; a.SD = SD; a.SH = SH; a.SW = SW; a.stride = stride;
    fill_tiles(a);
    return a.ntiles < 4 * cu_count() ? a.ntiles : 4 * cu_count();       // one partial per workgroup
}

hipError_t conv_c1_fwd_launch(ConvC1Args a, hipStream_t st) {
    fill_tiles(a);
    if (!conv_c1_supported(a.Cout, a.stride) || (int64_t)a.B * a.tiles_z * a.tiles_y * a.tiles_x > 0x7fffffff) return hipErrorInvalidValue;
    if (a.stride == 1) { if (a.Cout == 32) launch_c1(conv_c1_fwd_kernel<1, 1>, a.ntiles, a, st); else launch_c1(conv_c1_fwd_kernel<1, 2>, a.ntiles, a, st); }
    else { if (a.Cout == 32) launch_c1(conv_c1_fwd_kernel<2, 1>, a.ntiles, a, st); else launch_c1(conv_c1_fwd_kernel<2, 2>, a.ntiles, a, st); }
    return hipGetLastError();
}

hipError_t conv_c1_dw_launch(ConvC1Args a, float* dw, hipStream_t st) {
    fill_tiles(a);
    if (!conv_c1_supported(a.Cout, a.stride)) return hipErrorInvalidValue;
    const int parts = conv_c1_dw_parts(a.B, a.SD, a.SH, a.SW, a.stride), grid = parts;
    if (a.stride == 1) { if (a.Cout == 32) launch_c1(conv_c1_dw_kernel<1, 1>, grid, a, st); else launch_c1(conv_c1_dw_kernel<1, 2>, grid, a, st); }
    else { if (a.Cout == 32) launch_c1(conv_c1_dw_kernel<2, 1>, grid, a, st); else launch_c1(conv_c1_dw_kernel<2, 2>, grid, a, st); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = a.Cout * kTapPad;
    hipLaunchKernelGGL(conv_c1_dw_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, st, a.workspace, dw, n, parts);
    return hipGetLastError();
}

// a.dy: [B][D][H][W][Cout]; a.w2d: w^T as [32 taps][Cout] bf16 (rows 27.. zero); a.y: dx [B][D][H][W]; a.SD.. = the volume extent
hipError_t conv_c1_dx_launch(ConvC1Args a, hipStream_t st) {
    if (!conv_c1_supported(a.Cout, 1) || a.stride != 1) return hipErrorInvalidValue;
    a.OD = a.SD; a.OH = a.SH; a.OW = a.SW;
    a.tiles_x = (a.OW + kDX - 1) / kDX; a.tiles_y = (a.OH + kDY - 1) / kDY; a.tiles_z = (a.OD + kDZ - 1) / kDZ;
    const int64_t n = (int64_t)a.B * a.tiles_z * a.tiles_y * a.tiles_x;
    if (n > 0x7fffffff) return hipErrorInvalidValue;
    if (a.Cout == 32) hipLaunchKernelGGL(conv_c1_dx_kernel<1>, dim3((unsigned)n), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(conv_c1_dx_kernel<2>, dim3((unsigned)n), dim3(512), 0, st, a);
    return hipGetLastError();
}

bool conv3_halo_supported(int CI, int CO) { return (CI == 32 || CI == 64) && (CO == 32 || CO == 64); }

hipError_t conv3_halo_launch(Conv3Args a, hipStream_t st) {
    if (!conv3_halo_supported(a.CI, a.CO)) return hipErrorInvalidValue;
    a.tiles_x = (a.W + kBX - 1) / kBX; a.tiles_y = (a.H + kBY - 1) / kBY; a.tiles_z = (a.D + kBZ - 1) / kBZ;
    const int64_t n = (int64_t)a.B * a.tiles_z * a.tiles_y * a.tiles_x;
    if (n > 0x7fffffff) return hipErrorInvalidValue;
    const dim3 grid((unsigned)n), blk(256);
    if (a.CI == 64 && a.CO == 32) hipLaunchKernelGGL((conv3_halo_kernel<64, 32>), grid, blk, 0, st, a);
    else if (a.CI == 32 && a.CO == 64) hipLaunchKernelGGL((conv3_halo_kernel<32, 64>), grid, blk, 0, st, a);
    else if (a.CI == 32 && a.CO == 32) hipLaunchKernelGGL((conv3_halo_kernel<32, 32>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((conv3_halo_kernel<64, 64>), grid, blk, 0, st, a);
    return hipGetLastError();
}

bool conv_o1_supported(int C) { return C == 8 || C == 16 || C == 32 || C == 64 || C == 128; }

int conv_o1_bwd_blocks(int64_t M, int C) {
    const int64_t want = (M * (C / 8) + 256 * 4 - 1) / (256 * 4);
    const int cap = 8 * cu_count();
    return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}

hipError_t conv_o1_fwd_launch(const void* x, const void* w, const float* bias, void* y, int64_t M, int C, hipStream_t st) {
    if (!conv_o1_supported(C)) return hipErrorInvalidValue;
    const int64_t want = (M * (C / 8) + 256 * 4 - 1) / (256 * 4);
    const int grid = (int)(want < 16 * (int64_t)cu_count() ? (want < 1 ? 1 : want) : 16 * cu_count());
    const bf16 *xp = (const bf16*)x, *wp = (const bf16*)w;
    bf16* yp = (bf16*)y;
    switch (C / 8) {
        case 1: hipLaunchKernelGGL(conv_o1_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, xp, wp, bias, yp, M); break;
        case 2: hipLaunchKernelGGL(conv_o1_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, xp, wp, bias, yp, M); break;
        case 4: hipLaunchKernelGGL(conv_o1_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, xp, wp, bias, yp, M); break;
        case 8: hipLaunchKernelGGL(conv_o1_fwd_kernel<8>, dim3(grid), dim3(256), 0, st, xp, wp, bias, yp, M); break;
        default: hipLaunchKernelGGL(conv_o1_fwd_kernel<16>, dim3(grid), dim3(256), 0, st, xp, wp, bias, yp, M); break;
    }
    return hipGetLastError();
}

hipError_t conv_o1_bwd_launch(const void* x, const void* dy, const void* w, void* dx, float* dwb, float* workspace, int64_t M, int C, hipStream_t st) {
    if (!conv_o1_supported(C)) return hipErrorInvalidValue;
    const int grid = conv_o1_bwd_blocks(M, C);
    const bf16 *xp = (const bf16*)x, *dyp = (const bf16*)dy, *wp = (const bf16*)w;
    bf16* dxp = (bf16*)dx;
    switch (C / 8) {
        case 1: hipLaunchKernelGGL(conv_o1_bwd_kernel<1>, dim3(grid), dim3(256), 0, st, xp, dyp, wp, dxp, workspace, M); break;
        case 2: hipLaunchKernelGGL(conv_o1_bwd_kernel<2>, dim3(grid), dim3(256), 0, st, xp, dyp, wp, dxp, workspace, M); break;
        case 4: hipLaunchKernelGGL(conv_o1_bwd_kernel<4>, dim3(grid), dim3(256), 0, st, xp, dyp, wp, dxp, workspace, M); break;
        case 8: hipLaunchKernelGGL(conv_o1_bwd_kernel<8>, dim3(grid), dim3(256), 0, st, xp, dyp, wp, dxp, workspace, M); break;
        default: hipLaunchKernelGGL(conv_o1_bwd_kernel<16>, dim3(grid), dim3(256), 0, st, xp, dyp, wp, dxp, workspace, M); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = C + 1;
    hipLaunchKernelGGL(conv_c1_dw_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, st, workspace, dwb, n, grid);
    return hipGetLastError();
}

}  // namespace hvc
