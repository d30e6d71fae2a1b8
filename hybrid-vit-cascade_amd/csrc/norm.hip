// Normalisation layers of the two conv stems on channels-last activations x[B][P][C]
// (P = D*H*W positions), gfx950, HBM-bound:
//   GroupNorm(min(8,C), C) + SiLU            models/hybrid_vit_backbone.py:197-201
//   BatchNorm2d + ReLU (+ MaxPool2d)         models/diagnostic_losses.py:82-96
// Forward = one statistics pass (deterministic two-stage column reduction, fp64 finish) + one fused
// apply pass (normalise + affine + activation [+ pooling with recorded arg-max]).
// Backward = one reduction pass + one apply pass, the activation / pooling gradients recomputed on
// the fly from the saved conv output and the 1-byte arg-max map.
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

constexpr int kMaxC = 512;

// ---- generic per-channel partial sums over a slab of positions ------------------------------------
// Block (256 threads) = (256 / C8) position lanes x C8 = C/8 channel octets.  Fn maps the 8 loaded
// values of one position to two 8-vectors (u, v) whose per-channel sums are wanted.
template <typename F>
__device__ __forceinline__ void column_sums(int P, int C, int chunk, int nchunk, float* red /*[2][8 lanes][C]*/,
                                            float* out /*[2][C]*/, F&& body) {
    const int c8n = C / 8;
    const int lanes = 256 / c8n;                  // position lanes per block (C8 divides 256)
    const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
    const int per = (P + nchunk - 1) / nchunk;
    const int p0 = chunk * per, p1 = min(P, p0 + per);
    float su[8], sv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { su[j] = 0.f; sv[j] = 0.f; }
    if (pl < lanes) {
        for (int p = p0 + pl; p < p1; p += lanes) {
            float u[8], v[8];
            body(p, c8, u, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { su[j] += u[j]; sv[j] += v[j]; }
        }
    }
    // fold the position lanes in a fixed order, 8 at a time through LDS
    for (int base = 0; base < lanes; base += 8) {
        __syncthreads();
        if (pl >= base && pl < base + 8 && pl < lanes) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[((0 * 8 + (pl - base)) * C) + c8 * 8 + j] = su[j];
                red[((1 * 8 + (pl - base)) * C) + c8 * 8 + j] = sv[j];
            }
        }
        __syncthreads();
        const int n = min(8, lanes - base);
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            const int which = i / C, c = i % C;
            float s = base == 0 ? 0.f : out[i];
            for (int l = 0; l < n; ++l) s += red[((which * 8 + l) * C) + c];
            out[i] = s;
        }
    }
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    Chunk8<T> c = load_chunk<T>(p, 8, true);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = chunk_get<T>(c, j);
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(v[j]);
        *reinterpret_cast<bf16x8*>(p) = o;
    } else {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// ---- statistics: partial[b][chunk][2][C] = (sum x, sum x^2) ------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void chan_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, int P, int C, int nchunk) {
    __shared__ float red[2 * 8 * kMaxC];
    __shared__ float out[2 * kMaxC];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const T* xb = x + (int64_t)b * P * C;
    column_sums(P, C, chunk, nchunk, red, out, [&](int p, int c8, float (&u)[8], float (&v)[8]) {
        load8<T>(xb + (int64_t)p * C + c8 * 8, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = u[j] * u[j];
    });
    float* dst = partial + ((int64_t)b * nchunk + chunk) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = out[i];
}

// Sums partial[r][which][c] over r in [r0, r1) for 32 consecutive channels per block (256 threads = 32 channels x 8
// row groups, fp64, fixed order).  Result valid in the threads with (tid >> 5) == 0.
__device__ __forceinline__ void chan_sums2(const float* __restrict__ partial, int r0, int r1, int C, int c, bool valid,
                                           double (*red)[8][32], double& s0, double& s1) {
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    double a = 0.0, b = 0.0;
    if (valid)
        for (int r = r0 + rg; r < r1; r += 8) { a += partial[(int64_t)r * 2 * C + c]; b += partial[(int64_t)r * 2 * C + C + c]; }
    red[0][rg][cl] = a;
    red[1][rg][cl] = b;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s0 += red[0][k][cl]; s1 += red[1][k][cl]; }
}

// GroupNorm finish: stats[b][g] = (mean, rstd) over P * cg elements.  grid = B * G blocks of 256 threads.
__global__ __launch_bounds__(256) void gn_finish_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                         int B, int P, int C, int G, int nchunk, float eps) {
    __shared__ double red[2][256];
    const int i = blockIdx.x, b = i / G, g = i % G, cg = C / G;
    double s = 0.0, ss = 0.0;
    for (int j = threadIdx.x; j < nchunk * cg; j += 256) {
        const int ch = j / cg, c = g * cg + j % cg;
        const float* p = partial + ((int64_t)b * nchunk + ch) * 2 * C;
        s += p[c];
        ss += p[C + c];
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)P * cg, mean = red[0][0] / n;
        double var = red[1][0] / n - mean * mean;
        if (var < 0) var = 0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// BatchNorm finish (train): per channel over all samples; also updates the running statistics.  grid = ceil(C/32).
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ partial, float* __restrict__ stats /*[C][2]*/,
                                                         float* running_mean, float* running_var, int B, int P, int C, int nchunk,
                                                         float eps, float momentum) {
    __shared__ double red[2][8][32];
    const int c = 32 * blockIdx.x + (threadIdx.x & 31);
    const bool valid = c < C;
    double s, ss;
    chan_sums2(partial, 0, B * nchunk, C, c, valid, red, s, ss);
    if (!valid || (threadIdx.x >> 5) != 0) return;
    const double n = (double)B * P, mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0) var = 0;
    stats[2 * c] = (float)mean;
    stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n > 1 ? var * n / (n - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// BatchNorm eval: stats from the running buffers.
__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* rm, const float* rv, float* stats, int C, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    stats[2 * c] = rm[c];
    stats[2 * c + 1] = rsqrtf(rv[c] + eps);
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = 1.f / (1.f + __expf(-z));
    return s * (1.f + z * (1.f - s));
}
// act: 0 = SiLU (voxel-embed stem), 1 = GELU(erf) (cascade glue: model_progressive.py:37-51, 169-174)
__device__ __forceinline__ float act_f(float z, int act) {
    return act == 0 ? silu_f(z) : gelu_f(z);
}
__device__ __forceinline__ float act_grad_f(float z, int act) {
    return act == 0 ? silu_grad_f(z) : gelu_grad_f(z);
}

// ---- GroupNorm + SiLU apply ---------------------------------------------------------------------------
// grid = (blocks, B).  C / 8 divides 256, so a thread's 8 channels (tid % (C / 8)) are the same for every chunk it visits: affine
// parameters and - when its 8 channels lie in one group - the statistics are fetched once, and the sweep has no division in it
// (round 4: the flat 64-bit index cost two 64-bit divisions per 16-byte chunk, ~300 instructions; 908 -> ~480 us at 256^3 x 32).
template <typename T>
__global__ __launch_bounds__(256) void gn_silu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int P, int C, int G, int act) {
    const int c8n = C / 8, cg = C / G, b = blockIdx.y;
    const int c8 = threadIdx.x % c8n;
    const int64_t total = (int64_t)P * c8n, step = (int64_t)gridDim.x * 256;
    const T* xb = x + (int64_t)b * P * C;
    T* yb = y + (int64_t)b * P * C;
    float gam[8], bet[8], mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j, sg = b * G + c / cg;
        gam[j] = gamma[c]; bet[j] = beta[c];
        mean[j] = stats[2 * sg]; rstd[j] = stats[2 * sg + 1];
    }
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += 2 * step) {
        const bool two = idx + step < total;
        float v[8], w[8];
        load8<T>(xb + idx * 8, v);
        if (two) load8<T>(xb + (idx + step) * 8, w);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_f((v[j] - mean[j]) * rstd[j] * gam[j] + bet[j], act);
        store8<T>(yb + idx * 8, v);
        if (two) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = act_f((w[j] - mean[j]) * rstd[j] * gam[j] + bet[j], act);
            store8<T>(yb + (idx + step) * 8, w);
        }
    }
}

// backward reduction: partial[b][chunk][2][C] = (sum ds, sum ds * xhat), ds = dy * silu'(z)
template <typename T>
__global__ __launch_bounds__(256) void gn_silu_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ partial, int P, int C, int G, int nchunk, int act) {
    __shared__ float red[2 * 8 * kMaxC];
    __shared__ float out[2 * kMaxC];
    const int b = blockIdx.y, chunk = blockIdx.x, cg = C / G;
    const T* xb = x + (int64_t)b * P * C;
    const T* db = dy + (int64_t)b * P * C;
    // a thread keeps its 8 channels for the whole sweep (column_sums: c8 = tid % (C/8)): statistics and affine parameters of
    // those channels are fetched once, as xhat = x * rs + nm and z = xhat * gamma + beta
    float rs[8], nm[8], gam[8], bet[8];
    {
        const int c8 = threadIdx.x % (C / 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j, g = c / cg;
            rs[j] = stats[2 * (b * G + g) + 1];
            nm[j] = stats[2 * (b * G + g)];          // the mean itself: xhat = (x - mean) * rstd, the form the forward and dx kernels use
            gam[j] = gamma[c];
            bet[j] = beta[c];
        }
    }
    column_sums(P, C, chunk, nchunk, red, out, [&](int p, int c8, float (&u)[8], float (&v)[8]) {
        float xv[8], dv[8];
        load8<T>(xb + (int64_t)p * C + c8 * 8, xv);
        load8<T>(db + (int64_t)p * C + c8 * 8, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - nm[j]) * rs[j];      // not fma(x, rstd, -mean rstd): that cancels ~|mean| / std ulps when |mean| >> std
            const float ds = dv[j] * act_grad_f(fmaf(xh, gam[j], bet[j]), act);
            u[j] = ds;
            v[j] = ds * xh;
        }
    });
    float* dst = partial + ((int64_t)b * nchunk + chunk) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = out[i];
}

// finish: blocks [0, ceil(C/32)) -> dgamma, dbeta [C];  blocks ceil(C/32) + (b*G + g) -> gsum[b][g] = (sum_c gamma ds, sum_c gamma ds xhat)
__global__ __launch_bounds__(256) void gn_bwd_finish_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                             float* dgamma, float* dbeta, float* gsum, int B, int C, int G, int nchunk) {
    __shared__ double red[2][8][32];
    const int ncb = (C + 31) / 32;
    if ((int)blockIdx.x < ncb) {
        const int c = 32 * blockIdx.x + (threadIdx.x & 31);
        const bool valid = c < C;
        double s, sx;
        chan_sums2(partial, 0, B * nchunk, C, c, valid, red, s, sx);
        if (valid && (threadIdx.x >> 5) == 0) { dbeta[c] = (float)s; dgamma[c] = (float)sx; }
        return;
    }
    double* flat0 = &red[0][0][0];
    double* flat1 = &red[1][0][0];
    const int j = blockIdx.x - ncb, b = j / G, g = j % G, cg = C / G;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < nchunk * cg; k += 256) {
        const int ch = k / cg, c = g * cg + k % cg;
        const float* p = partial + ((int64_t)b * nchunk + ch) * 2 * C;
        s1 += (double)gamma[c] * p[c];
        s2 += (double)gamma[c] * p[C + c];
    }
    flat0[threadIdx.x] = s1;
    flat1[threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { flat0[threadIdx.x] += flat0[threadIdx.x + off]; flat1[threadIdx.x] += flat1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { gsum[2 * j] = (float)flat0[0]; gsum[2 * j + 1] = (float)flat1[0]; }
}

// grid = (blocks, B); a thread keeps its 8 channels, as in gn_silu_fwd_kernel
template <typename T>
__global__ __launch_bounds__(256) void gn_silu_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                                 const float* __restrict__ stats, const float* __restrict__ gsum,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 int P, int C, int G, int act) {
    const int c8n = C / 8, cg = C / G, b = blockIdx.y;
    const int c8 = threadIdx.x % c8n;
    const float inv_n = 1.f / ((float)P * cg);
    const int64_t total = (int64_t)P * c8n, step = (int64_t)gridDim.x * 256;
    const T* xb = x + (int64_t)b * P * C;
    const T* db = dy + (int64_t)b * P * C;
    T* ob = dx + (int64_t)b * P * C;
    float gam[8], bet[8], mean[8], rstd[8], s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j, sg = b * G + c / cg;
        gam[j] = gamma[c]; bet[j] = beta[c];
        mean[j] = stats[2 * sg]; rstd[j] = stats[2 * sg + 1];
        s0[j] = gsum[2 * sg]; s1[j] = gsum[2 * sg + 1];
    }
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += step) {
        float xv[8], dv[8], o[8];
        load8<T>(xb + idx * 8, xv);
        load8<T>(db + idx * 8, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - mean[j]) * rstd[j];
            const float ds = dv[j] * act_grad_f(xh * gam[j] + bet[j], act);
            o[j] = rstd[j] * (ds * gam[j] - (s0[j] + xh * s1[j]) * inv_n);
        }
        store8<T>(ob + idx * 8, o);
    }
}

// ---- BatchNorm + ReLU + MaxPool (2-D, channels-last [N][H][W][C]) ------------------------------------
// pooled[n][hp][wp][c] = max over the k x k window (stride s, pad p) of relu(bn(x)); amax = window-local
// index of the first maximum (row-major scan, as ATen), 255 for a window with no valid cell.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ amax,
                                                                const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const PoolGeom pg) {
    const int c8n = pg.C / 8;
    const int64_t total = (int64_t)pg.N * pg.HP * pg.WP * c8n;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(idx % c8n);
        int64_t t = idx / c8n;
        const int wp = (int)(t % pg.WP); t /= pg.WP;
        const int hp = (int)(t % pg.HP);
        const int n = (int)(t / pg.HP);
        float a[8], sh[8], best[8];
        int bi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j;
            a[j] = stats[2 * c + 1] * gamma[c];
            sh[j] = beta[c] - stats[2 * c] * a[j];
            best[j] = -INFINITY;
            bi[j] = 255;
        }
        for (int kh = 0; kh < pg.k; ++kh) {
            const int h = hp * pg.s + kh - pg.p;
            if (h < 0 || h >= pg.H) continue;
            for (int kw = 0; kw < pg.k; ++kw) {
                const int w = wp * pg.s + kw - pg.p;
                if (w < 0 || w >= pg.W) continue;
                float v[8];
                load8<T>(x + (((int64_t)n * pg.H + h) * pg.W + w) * pg.C + c8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = fmaxf(v[j] * a[j] + sh[j], 0.f);
                    if (z > best[j]) { best[j] = z; bi[j] = kh * pg.k + kw; }
                }
            }
        }
        store8<T>(y + idx * 8, best);
        if (amax) {
            uint8_t* ap = amax + idx * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) ap[j] = (uint8_t)bi[j];
        }
    }
}

// gradient reaching relu(bn(x)) at (n,h,w,c8) through the pooling: sum of dpool over the windows whose
// recorded arg-max is this cell; zero where bn(x) <= 0.
template <typename T>
__device__ __forceinline__ void pool_grad_gather(const T* __restrict__ dpool, const uint8_t* __restrict__ amax, const PoolGeom& pg,
                                                 int n, int h, int w, int c8, float (&g)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    if (!amax) {       // no pooling: identity
        load8<T>(dpool + (((int64_t)n * pg.H + h) * pg.W + w) * pg.C + c8 * 8, g);
        return;
    }
    for (int kh = 0; kh < pg.k; ++kh) {
        const int nh = h + pg.p - kh;
        if (nh < 0 || nh % pg.s) continue;
        const int hp = nh / pg.s;
        if (hp >= pg.HP) continue;
        for (int kw = 0; kw < pg.k; ++kw) {
            const int nw = w + pg.p - kw;
            if (nw < 0 || nw % pg.s) continue;
            const int wp = nw / pg.s;
            if (wp >= pg.WP) continue;
            const int64_t o = (((int64_t)n * pg.HP + hp) * pg.WP + wp) * pg.C + c8 * 8;
            float d[8];
            load8<T>(dpool + o, d);
            const uint8_t* ap = amax + o;
            const int me = kh * pg.k + kw;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (ap[j] == me) g[j] += d[j];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dpool,
                                                                       const uint8_t* __restrict__ amax, const float* __restrict__ stats,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       float* __restrict__ partial, const PoolGeom pg, int nchunk) {
    __shared__ float red[2 * 8 * kMaxC];
    __shared__ float out[2 * kMaxC];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int P = pg.H * pg.W, C = pg.C;
    column_sums(P, C, chunk, nchunk, red, out, [&](int p, int c8, float (&u)[8], float (&v)[8]) {
        const int h = p / pg.W, w = p % pg.W;
        float xv[8], g[8];
        load8<T>(x + ((int64_t)n * P + p) * C + c8 * 8, xv);
        pool_grad_gather<T>(dpool, amax, pg, n, h, w, c8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j;
            const float xh = (xv[j] - stats[2 * c]) * stats[2 * c + 1];
            const float ds = (xh * gamma[c] + beta[c] > 0.f) ? g[j] : 0.f;
            u[j] = ds;
            v[j] = ds * xh;
        }
    });
    float* dst = partial + ((int64_t)n * nchunk + chunk) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = out[i];
}

// finish: dgamma, dbeta and the per-channel means (c1, c2) of ds and ds * xhat.  grid = ceil(C/32).
__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float* __restrict__ partial, float* dgamma, float* dbeta, float* cmean,
                                                             int N, int P, int C, int nchunk) {
    __shared__ double red[2][8][32];
    const int c = 32 * blockIdx.x + (threadIdx.x & 31);
    const bool valid = c < C;
    double s, sx;
    chan_sums2(partial, 0, N * nchunk, C, c, valid, red, s, sx);
    if (!valid || (threadIdx.x >> 5) != 0) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)sx;
    const double n = (double)N * P;
    cmean[2 * c] = (float)(s / n);
    cmean[2 * c + 1] = (float)(sx / n);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dpool,
                                                                      const uint8_t* __restrict__ amax, T* __restrict__ dx,
                                                                      const float* __restrict__ stats, const float* __restrict__ cmean,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const PoolGeom pg, int training) {
    const int c8n = pg.C / 8;
    const int64_t total = (int64_t)pg.N * pg.H * pg.W * c8n;
    // The grid stride is a multiple of C / 8 for the stems' power-of-two channel counts, so a thread keeps its 8 channels
    // for the whole sweep and their coefficients are fetched once (otherwise they are re-fetched per element).
    const bool fixed = ((int64_t)gridDim.x * 256) % c8n == 0;
    float mean[8], rstd[8], gam[8], bet[8], m1[8], m2[8];
    auto coeffs = [&](int c8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j;
            mean[j] = stats[2 * c]; rstd[j] = stats[2 * c + 1];
            gam[j] = gamma[c]; bet[j] = beta[c];
            m1[j] = training ? cmean[2 * c] : 0.f; m2[j] = training ? cmean[2 * c + 1] : 0.f;
        }
    };
    if (fixed) coeffs((int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % c8n));
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(idx % c8n);
        int64_t t = idx / c8n;
        const int w = (int)(t % pg.W); t /= pg.W;
        const int h = (int)(t % pg.H);
        const int n = (int)(t / pg.H);
        if (!fixed) coeffs(c8);
        float xv[8], g[8], o[8];
        load8<T>(x + idx * 8, xv);
        pool_grad_gather<T>(dpool, amax, pg, n, h, w, c8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - mean[j]) * rstd[j];
            const float ds = (xh * gam[j] + bet[j] > 0.f) ? g[j] : 0.f;
            o[j] = gam[j] * rstd[j] * (ds - m1[j] - xh * m2[j]);      // eval mode: m1 = m2 = 0
        }
        store8<T>(dx + idx * 8, o);
    }
}

int grid_for(int64_t work) {
    int64_t blocks = (work + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

// Position slabs per image for the per-channel reductions.  Up to 1024: at batch 1 (cascade stage 3, 256^3 x 32 channels) the
// old cap of 128 slabs put 128 workgroups on 256 CUs, one per CU - the reduce passes ran at half the chip with nothing to
// hide their latency (gn_silu_bwd_reduce 1240 us, 1.7 TB/s); >= 2048 positions per slab keep the LDS fold negligible.
int norm_chunks(int P) {
    const int small = (P + 255) / 256 < 128 ? (P + 255) / 256 : 128;      // up to 32768 positions: 256 per slab; then 128 slabs ...
    const int large = (P + 2047) / 2048 < 1024 ? (P + 2047) / 2048 : 1024; // ... until 2048 per slab gives more, up to 1024 slabs
    const int n = small > large ? small : large;
    return n < 1 ? 1 : n;
}

#define HVC_DISPATCH_T(is_bf16, CALL)            \
    do {                                         \
        if (is_bf16) { using T = bf16; CALL; }   \
        else { using T = float; CALL; }          \
    } while (0)

// blocks per sample of the GroupNorm apply kernels: the chip's share of grid_for()'s cap, at least one
static int apply_blocks(const NormArgs& a) {
    const int64_t want = ((int64_t)a.P * a.C / 8 + 255) / 256;
    const int64_t cap = (grid_for((int64_t)1 << 40) + a.B - 1) / a.B;
    return (int)(want < cap ? (want < 1 ? 1 : want) : (cap < 1 ? 1 : cap));
}

hipError_t groupnorm_silu_fwd_launch(const NormArgs& a, hipStream_t st) {
    if (a.C % 8 || a.C > kMaxC || 256 % (a.C / 8) || a.C % a.G) return hipErrorInvalidValue;
    const int nch = norm_chunks(a.P);
    HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(chan_stats_kernel<T>, dim3(nch, a.B), dim3(256), 0, st, (const T*)a.x, a.partial, a.P, a.C, nch));
    hipLaunchKernelGGL(gn_finish_kernel, dim3(a.B * a.G), dim3(256), 0, st, a.partial, a.stats, a.B, a.P, a.C, a.G, nch, a.eps);
    HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(gn_silu_fwd_kernel<T>, dim3(apply_blocks(a), a.B), dim3(256), 0, st,
                                                 (const T*)a.x, (T*)a.y, a.stats, a.gamma, a.beta, a.P, a.C, a.G, a.act));
    return hipGetLastError();
}

hipError_t groupnorm_silu_bwd_launch(const NormArgs& a, hipStream_t st) {
    if (a.C % 8 || a.C > kMaxC || 256 % (a.C / 8) || a.C % a.G) return hipErrorInvalidValue;
    const int nch = norm_chunks(a.P);
    HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(gn_silu_bwd_reduce_kernel<T>, dim3(nch, a.B), dim3(256), 0, st, (const T*)a.x, (const T*)a.dy,
                                                 a.stats, a.gamma, a.beta, a.partial, a.P, a.C, a.G, nch, a.act));
    hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3((a.C + 31) / 32 + a.B * a.G), dim3(256), 0, st, a.partial, a.gamma, a.dgamma, a.dbeta,
                       a.gsum, a.B, a.C, a.G, nch);
    HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(gn_silu_bwd_apply_kernel<T>, dim3(apply_blocks(a), a.B), dim3(256), 0, st,
                                                 (const T*)a.x, (const T*)a.dy, (T*)a.dx, a.stats, a.gsum, a.gamma, a.beta, a.P, a.C, a.G, a.act));
    return hipGetLastError();
}

hipError_t bn_relu_pool_fwd_launch(const NormArgs& a, const PoolGeom& pg, hipStream_t st) {
    const int C = pg.C, P = pg.H * pg.W;
    if (C % 8 || C > kMaxC || 256 % (C / 8)) return hipErrorInvalidValue;
    const int nch = norm_chunks(P);
    if (a.training) {
        HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(chan_stats_kernel<T>, dim3(nch, pg.N), dim3(256), 0, st, (const T*)a.x, a.partial, P, C, nch));
        hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 31) / 32), dim3(256), 0, st, a.partial, a.stats, a.running_mean, a.running_var,
                           pg.N, P, C, nch, a.eps, a.momentum);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, a.running_mean, a.running_var, a.stats, C, a.eps);
    }
    HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<T>, dim3(grid_for((int64_t)pg.N * pg.HP * pg.WP * C / 8)), dim3(256), 0, st,
                                                 (const T*)a.x, (T*)a.y, a.amax, a.stats, a.gamma, a.beta, pg));
    return hipGetLastError();
}

hipError_t bn_relu_pool_bwd_launch(const NormArgs& a, const PoolGeom& pg, hipStream_t st) {
    const int C = pg.C, P = pg.H * pg.W;
    if (C % 8 || C > kMaxC || 256 % (C / 8)) return hipErrorInvalidValue;
    const int nch = norm_chunks(P);
    HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(bn_relu_pool_bwd_reduce_kernel<T>, dim3(nch, pg.N), dim3(256), 0, st, (const T*)a.x, (const T*)a.dy,
                                                 a.amax, a.stats, a.gamma, a.beta, a.partial, pg, nch));
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 31) / 32), dim3(256), 0, st, a.partial, a.dgamma, a.dbeta, a.gsum, pg.N, P, C, nch);
    HVC_DISPATCH_T(a.is_bf16, hipLaunchKernelGGL(bn_relu_pool_bwd_apply_kernel<T>, dim3(grid_for((int64_t)pg.N * P * C / 8)), dim3(256), 0, st,
                                                 (const T*)a.x, (const T*)a.dy, a.amax, (T*)a.dx, a.stats, a.gsum, a.gamma, a.beta, pg, a.training));
    return hipGetLastError();
}

}  // namespace hvc
