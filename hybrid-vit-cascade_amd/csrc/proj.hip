// Small fused epilogues around the projection losses and the X-ray stem (gfx950, HBM / latency bound):
//
//  * resize_loss: bilinear resize of a projection (B, h, w) to the X-ray raster (S1, S2) fused with the L1 / MSE
//    reduction against the target X-ray - the tail of
//      direct_regression/progressive_cascade/loss_multiscale.py:269-293  (DRRReprojectionLoss: align_corners=False, L1)
//      models/diagnostic_losses.py:161-169                               (ProjectionLoss: align_corners=True, MSE)
//    Forward: one pass, the resized image is never written.  Backward: d loss / d resized is formed in one pass and handed
//    to the adjoint of the resize (the separable, atomic-free adjoint of spatial.hip with depth 1).
//  * view_mean_gap: mean over the V views of the channels-last X-ray feature maps (B*V, P, E) fused with the global
//    average pool over the P = H'*W' positions (models/diagnostic_losses.py:126, :131).
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

struct AxisMap2 {       // ATen area_pixel_compute_source_index (as spatial.hip)
    float r;
    int ac;
    __device__ __forceinline__ float src(int o) const { return ac ? r * o : fmaxf(r * (o + 0.5f) - 0.5f, 0.f); }
};
__device__ __forceinline__ AxisMap2 axis_map2(int in, int out, int ac) {
    AxisMap2 m;
    m.ac = ac;
    m.r = ac ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f) : (float)in / (float)out;
    return m;
}

__device__ __forceinline__ float bilinear_at(const float* __restrict__ img, int h, int w, const AxisMap2& mh, const AxisMap2& mw, int i, int j) {
    const float fy = mh.src(i), fx = mw.src(j);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float top = img[(int64_t)y0 * w + x0] * (1.f - lx) + img[(int64_t)y0 * w + x1] * lx;
    const float bot = img[(int64_t)y1 * w + x0] * (1.f - lx) + img[(int64_t)y1 * w + x1] * lx;
    return top * (1.f - ly) + bot * ly;
}

// partial[blk] = sum over the block's output pixels of |r - t| (mode 0) or (r - t)^2 (mode 1)
__global__ __launch_bounds__(256) void resize_loss_fwd_kernel(const float* __restrict__ proj, const float* __restrict__ target, float* __restrict__ partial,
                                                              int B, int h, int w, int S1, int S2, int64_t tb, int ac, int mode) {
    __shared__ float red[4];
    const AxisMap2 mh = axis_map2(h, S1, ac), mw = axis_map2(w, S2, ac);
    const int64_t total = (int64_t)B * S1 * S2;
    float acc = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx % S2), i = (int)((idx / S2) % S1), b = (int)(idx / ((int64_t)S1 * S2));
        const float d = bilinear_at(proj + (int64_t)b * h * w, h, w, mh, mw, i, j) - target[(int64_t)b * tb + (int64_t)i * S2 + j];
        acc += mode ? d * d : fabsf(d);
    }
    const float rsum = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = rsum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void scalar_finish_kernel(const float* __restrict__ partial, int nblk, float* out, double scale) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

// dres[b][i][j] = gscale[0] / N * phi'(r - t): sign (L1; 0 at a tie, as torch) or 2 (r - t) (MSE)
__global__ __launch_bounds__(256) void resize_loss_grad_kernel(const float* __restrict__ proj, const float* __restrict__ target, const float* __restrict__ gscale,
                                                               float* __restrict__ dres, int B, int h, int w, int S1, int S2, int64_t tb, int ac, int mode,
                                                               float inv_n) {
    const AxisMap2 mh = axis_map2(h, S1, ac), mw = axis_map2(w, S2, ac);
    const int64_t total = (int64_t)B * S1 * S2;
    const float g = gscale[0] * inv_n;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx % S2), i = (int)((idx / S2) % S1), b = (int)(idx / ((int64_t)S1 * S2));
        const float d = bilinear_at(proj + (int64_t)b * h * w, h, w, mh, mw, i, j) - target[(int64_t)b * tb + (int64_t)i * S2 + j];
        dres[idx] = mode ? 2.f * d * g : (d > 0.f ? g : (d < 0.f ? -g : 0.f));
    }
}

// ---- mean over views + global average pool, channels-last f[(b * V + v)][p][e] ------------------------------------------
// grid (chunks of positions, B): thread -> 8 channels x a position lane; mean[b][p][e] written, partial[b][chunk][e] = sum over
// the chunk's positions (finished in a fixed order by view_gap_finish_kernel).
template <typename T>
__global__ __launch_bounds__(256) void view_mean_gap_fwd_kernel(const T* __restrict__ f, float* __restrict__ mean, float* __restrict__ partial,
                                                                int V, int P, int E, int chunk) {
    extern __shared__ float sm[];                 // [256 / (E/8)][E]
    const int e8n = E / 8, lanes_p = 256 / e8n;
    const int e8 = threadIdx.x % e8n, pl = threadIdx.x / e8n;
    const int b = blockIdx.y, p0 = blockIdx.x * chunk;
    const float inv_v = 1.f / (float)V;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + pl; p < min(P, p0 + chunk); p += lanes_p) {
        float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int v = 0; v < V; ++v) {
            const T* src = f + (((int64_t)b * V + v) * P + p) * E + e8 * 8;
            Chunk8<T> c = load_chunk<T>(src, 8, true);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] += chunk_get<T>(c, j);
        }
        float* dst = mean + ((int64_t)b * P + p) * E + e8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] *= inv_v; acc[j] += m[j]; }
        *reinterpret_cast<f32x4*>(dst) = f32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{m[4], m[5], m[6], m[7]};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sm[pl * E + e8 * 8 + j] = acc[j];
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
        float s = 0.f;
        for (int q = 0; q < lanes_p; ++q) s += sm[q * E + e];
        partial[((int64_t)b * gridDim.x + blockIdx.x) * E + e] = s;
    }
}

__global__ __launch_bounds__(256) void view_gap_finish_kernel(const float* __restrict__ partial, float* __restrict__ pooled, int nchunk, int E, float inv_p) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < E; e += 256) {
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) s += partial[((int64_t)b * nchunk + c) * E + e];
        pooled[(int64_t)b * E + e] = s * inv_p;
    }
}

// df[(b, v)][p][e] = (dmean[b][p][e] + dpooled[b][e] / P) / V   for every view v
template <typename T>
__global__ __launch_bounds__(256) void view_mean_gap_bwd_kernel(const float* __restrict__ dmean, const float* __restrict__ dpooled, T* __restrict__ df,
                                                                int B, int V, int P, int E) {
    const int e8n = E / 8;
    const int64_t total = (int64_t)B * P * e8n;
    const float inv_v = 1.f / (float)V, inv_p = 1.f / (float)P;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int e8 = (int)(idx % e8n);
        const int64_t bp = idx / e8n;
        const int b = (int)(bp / P);
        const int p = (int)(bp % P);
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = dpooled ? dpooled[(int64_t)b * E + e8 * 8 + j] * inv_p : 0.f;
            if (dmean) d += dmean[bp * E + e8 * 8 + j];
            g[j] = d * inv_v;
        }
        for (int v = 0; v < V; ++v) {
            T* dst = df + (((int64_t)b * V + v) * P + p) * E + e8 * 8;
            if constexpr (sizeof(T) == 2) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(g[j]);
                *reinterpret_cast<bf16x8*>(dst) = o;
            } else {
                *reinterpret_cast<f32x4*>(dst) = f32x4{g[0], g[1], g[2], g[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{g[4], g[5], g[6], g[7]};
            }
        }
    }
}

int blocks_for(int64_t work) {
    int64_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

int resize_loss_blocks(int B, int S1, int S2) { return blocks_for((int64_t)B * S1 * S2); }

hipError_t resize_loss_fwd_launch(const ResizeLossArgs& a, hipStream_t st) {
    const int nblk = resize_loss_blocks(a.B, a.S1, a.S2);
    hipLaunchKernelGGL(resize_loss_fwd_kernel, dim3(nblk), dim3(256), 0, st, a.proj, a.target, a.workspace, a.B, a.h, a.w, a.S1, a.S2, a.target_bstride,
                       a.align_corners, a.mode);
    hipLaunchKernelGGL(scalar_finish_kernel, dim3(1), dim3(256), 0, st, a.workspace, nblk, a.out, 1.0 / ((double)a.B * a.S1 * a.S2));
    return hipGetLastError();
}

hipError_t resize_loss_grad_launch(const ResizeLossArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(resize_loss_grad_kernel, dim3(resize_loss_blocks(a.B, a.S1, a.S2)), dim3(256), 0, st, a.proj, a.target, a.gscale, a.dres, a.B, a.h,
                       a.w, a.S1, a.S2, a.target_bstride, a.align_corners, a.mode, (float)(1.0 / ((double)a.B * a.S1 * a.S2)));
    return hipGetLastError();
}

int view_gap_chunks(int P) {
    int c = (P + 63) / 64;          // >= 64 positions per workgroup
    return c < 1 ? 1 : (c > 64 ? 64 : c);
}

hipError_t view_mean_gap_fwd_launch(const ViewGapArgs& a, hipStream_t st) {
    const int nchunk = view_gap_chunks(a.P);
    const int chunk = (a.P + nchunk - 1) / nchunk;
    const size_t lds = (size_t)(256 / (a.E / 8)) * a.E * sizeof(float);
    if (a.is_bf16) hipLaunchKernelGGL((view_mean_gap_fwd_kernel<bf16>), dim3(nchunk, a.B), dim3(256), lds, st, (const bf16*)a.f, a.mean, a.workspace, a.V, a.P, a.E, chunk);
    else hipLaunchKernelGGL((view_mean_gap_fwd_kernel<float>), dim3(nchunk, a.B), dim3(256), lds, st, (const float*)a.f, a.mean, a.workspace, a.V, a.P, a.E, chunk);
    hipLaunchKernelGGL(view_gap_finish_kernel, dim3(a.B), dim3(256), 0, st, a.workspace, a.pooled, nchunk, a.E, 1.f / (float)a.P);
    return hipGetLastError();
}

hipError_t view_mean_gap_bwd_launch(const ViewGapArgs& a, hipStream_t st) {
    const int64_t work = (int64_t)a.B * a.P * (a.E / 8);
    if (a.is_bf16) hipLaunchKernelGGL((view_mean_gap_bwd_kernel<bf16>), dim3(blocks_for(work)), dim3(256), 0, st, a.dmean, a.dpooled, (bf16*)a.df, a.B, a.V, a.P, a.E);
    else hipLaunchKernelGGL((view_mean_gap_bwd_kernel<float>), dim3(blocks_for(work)), dim3(256), 0, st, a.dmean, a.dpooled, (float*)a.df, a.B, a.V, a.P, a.E);
    return hipGetLastError();
}

}  // namespace hvc
