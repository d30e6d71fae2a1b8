// Row-streaming helpers around the residual branches (HBM-bound, deterministic two-stage
// column reductions):
//   branch_bwd : backward of   x_out = x + gate_b * z     (models/hybrid_vit_backbone.py:123,128,139)
//                dz = gate_b * dy * dropmask (cast to the GEMM dtype), dgate_b = sum_rows dy * z,
//                dbias = sum_rows dz   (the bias gradient of the Linear that produced z)
//   colsum     : bias gradient of a Linear whose output gradient is already materialised.
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

// 8 consecutive values <-> floats with one 16-byte (bf16) / two 16-byte (fp32) accesses
template <typename T>
__device__ __forceinline__ void ld8(const T* p, float (&v)[8]) {
    Chunk8<T> c = load_chunk<T>(p, 8, true);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = chunk_get<T>(c, j);
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const float (&v)[8]) {
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

// Fold the per-thread column partials of one block: thread (rl, c8) holds 8 columns of row-lane rl; result
// out[c] = sum over row lanes in a fixed order.  red: [rlanes][N] floats.
__device__ __forceinline__ void fold_rows(const float (&acc)[8], int rl, int c8, int rlanes, int N, float* red, float* out, bool active) {
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(size_t)rl * N + c8 * 8 + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
        float s = 0.f;
        for (int l = 0; l < rlanes; ++l) s += red[(size_t)l * N + c];
        out[c] = s;
    }
    __syncthreads();
}

// grid = nbatch * bpb blocks.  VEC (N % 8 == 0, N <= 2048): thread = (row lane, 8-column chunk), 16/32-byte accesses.
template <typename TO, bool VEC>
__global__ __launch_bounds__(256) void branch_bwd_kernel(const BranchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int bpb = a.blocks_per_batch;
    const int bidx = blockIdx.x / bpb, blk = blockIdx.x % bpb;
    const int rpb = a.rows_per_batch;
    const int per = (rpb + bpb - 1) / bpb;
    const int r0 = blk * per, r1 = min(rpb, r0 + per);
    const TO* z = reinterpret_cast<const TO*>(a.z);
    TO* dz = reinterpret_cast<TO*>(a.dz);
    float* pg = a.partial + ((size_t)blockIdx.x * 2 + 0) * a.N;
    float* pb = a.partial + ((size_t)blockIdx.x * 2 + 1) * a.N;
    if constexpr (VEC) {
        const int c8n = a.N / 8;
        const int rlanes = max(1, 256 / c8n);
        const int c8 = threadIdx.x % c8n, rl = threadIdx.x / c8n;
        const bool active = rl < rlanes && threadIdx.x < rlanes * c8n;
        float gt[8], sg[8], sb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gt[j] = 1.f; sg[j] = 0.f; sb[j] = 0.f; }
        if (active) {
            if (a.gate) ld8<float>(a.gate + (int64_t)bidx * a.N + c8 * 8, gt);
            for (int rr = r0 + rl; rr < r1; rr += rlanes) {
                const int64_t e = ((int64_t)bidx * rpb + rr) * a.N + c8 * 8;
                float d[8], v[8];
                ld8<float>(a.dy + e, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = d[j] * gt[j];
                if (a.drop_thresh) {
                    bool keep[8];
                    drop2d_keep8(drop2d_rowkey(seed_with_counter(a.seed_lo, a.seed_ctr), a.seed_hi, (uint64_t)bidx * rpb + rr), (uint32_t)(c8 * 8), a.drop_thresh, keep);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = keep[j] ? v[j] * a.keep_scale : 0.f;
                }
                if (z) {
                    float zz[8];
                    ld8<TO>(z + e, zz);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sg[j] += d[j] * zz[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) sb[j] += v[j];
                st8<TO>(dz + e, v);
            }
        }
        fold_rows(sg, rl, c8, rlanes, a.N, red, pg, active);
        fold_rows(sb, rl, c8, rlanes, a.N, red, pb, active);
    } else {
        for (int c = threadIdx.x; c < a.N; c += 256) {
            const float gt = a.gate ? a.gate[(int64_t)bidx * a.N + c] : 1.f;
            float sg = 0.f, sb = 0.f;
            for (int rr = r0; rr < r1; ++rr) {
                const int64_t e = ((int64_t)bidx * rpb + rr) * a.N + c;
                const float d = a.dy[e];
                float v = d * gt;
                if (a.drop_thresh)
                    v = drop2d_keep(drop2d_rowkey(seed_with_counter(a.seed_lo, a.seed_ctr), a.seed_hi, (uint64_t)bidx * rpb + rr), (uint32_t)c, a.drop_thresh) ? v * a.keep_scale : 0.f;
                if (z) sg += d * to_f<TO>(z[e]);
                sb += v;
                dz[e] = from_f<TO>(v);
            }
            pg[c] = sg;
            pb[c] = sb;
        }
    }
}

// grid (ceil(N/8), 1 + nbatch): y = 0 dbias over all blocks; y = 1 + b: dgate of sample b.
__global__ __launch_bounds__(256) void branch_final_kernel(const BranchArgs a, int nbatch) {
    __shared__ float red[32][kFinalCols];
    const int N = a.N, bpb = a.blocks_per_batch;
    const int c = kFinalCols * blockIdx.x + (threadIdx.x & 7);
    const bool valid = c < N;
    if (blockIdx.y == 0) {
        if (!a.dbias) return;
        const float s = block_colsum8(a.partial + N, nbatch * bpb, 2 * (int64_t)N, c, valid, red);
        if (valid && (threadIdx.x >> 3) == 0) a.dbias[c] = s;
    } else {
        if (!a.dgate) return;
        const int b = blockIdx.y - 1;
        const float s = block_colsum8(a.partial + (size_t)b * bpb * 2 * N, bpb, 2 * (int64_t)N, c, valid, red);
        if (valid && (threadIdx.x >> 3) == 0) a.dgate[(size_t)b * N + c] = s;
    }
}

// grid.x = row blocks.  VEC: thread = (row lane, 8-column chunk).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, float* partial, int M, int N, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int per = (M + nblk - 1) / nblk;
    const int r0 = blockIdx.x * per, r1 = min(M, r0 + per);
    float* out = partial + (size_t)blockIdx.x * N;
    if constexpr (VEC) {
        const int c8n = N / 8;
        const int rlanes = max(1, 256 / c8n);
        const int c8 = threadIdx.x % c8n, rl = threadIdx.x / c8n;
        const bool active = threadIdx.x < rlanes * c8n;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (active)
            for (int r = r0 + rl; r < r1; r += rlanes) {
                float v[8];
                ld8<T>(x + (int64_t)r * N + c8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        fold_rows(acc, rl, c8, rlanes, N, red, out, active);
    } else {
        // ragged column counts (the 1-channel output of model_progressive.py:123 gives N = 1): 256 / N rows in flight per
        // sweep instead of one thread walking a whole column, folded through LDS in a fixed order
        if (N <= 256) {
            const int rlanes = 256 / N;
            const int c = threadIdx.x % N, rl = threadIdx.x / N;
            float s = 0.f;
            if (rl < rlanes)
                for (int r = r0 + rl; r < r1; r += rlanes) s += to_f<T>(x[(int64_t)r * N + c]);
            if (rl < rlanes) red[rl * N + c] = s;
            __syncthreads();
            if (threadIdx.x < N) {
                float tot = 0.f;
                for (int k = 0; k < rlanes; ++k) tot += red[k * N + threadIdx.x];
                out[threadIdx.x] = tot;
            }
        } else {
            for (int c = threadIdx.x; c < N; c += 256) {
                float s = 0.f;
                for (int r = r0; r < r1; ++r) s += to_f<T>(x[(int64_t)r * N + c]);
                out[c] = s;
            }
        }
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, float* out, int N, int nblk) {
    __shared__ float red[32][kFinalCols];
    const int c = kFinalCols * blockIdx.x + (threadIdx.x & 7);
    const float s = block_colsum8(partial, nblk, (int64_t)N, c, c < N, red);
    if (c < N && (threadIdx.x >> 3) == 0) out[c] = s;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* x, TO* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = from_f<TO>(to_f<TI>(x[i]));
}

}  // namespace

int rowops_blocks(int rows) {
    int b = (rows + 31) / 32;
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return b;
}

hipError_t branch_bwd_launch(const BranchArgs& a, hipStream_t st) {
    if (a.rows % a.rows_per_batch != 0) return hipErrorInvalidValue;
    const int nbatch = a.rows / a.rows_per_batch;
    dim3 grid((unsigned)(nbatch * a.blocks_per_batch)), blk(256);
    const bool vec = (a.N % 8 == 0) && a.N <= 2048 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.dz) & 15) == 0 &&
                     (!a.z || (reinterpret_cast<uintptr_t>(a.z) & 15) == 0) && (!a.gate || (reinterpret_cast<uintptr_t>(a.gate) & 15) == 0);
    const size_t lds = vec ? (size_t)max(1, 256 / (a.N / 8)) * a.N * sizeof(float) : 0;
    if (a.out_bf16) {
        if (vec) hipLaunchKernelGGL((branch_bwd_kernel<bf16, true>), grid, blk, lds, st, a);
        else hipLaunchKernelGGL((branch_bwd_kernel<bf16, false>), grid, blk, 0, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((branch_bwd_kernel<float, true>), grid, blk, lds, st, a);
        else hipLaunchKernelGGL((branch_bwd_kernel<float, false>), grid, blk, 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (!a.dbias && !a.dgate) return hipSuccess;
    hipLaunchKernelGGL(branch_final_kernel, dim3((a.N + kFinalCols - 1) / kFinalCols, 1 + (a.dgate ? nbatch : 0)), blk, 0, st, a, nbatch);
    return hipGetLastError();
}

hipError_t colsum_launch(const void* x, float* partial, float* out, int M, int N, int nblk, int is_bf16, hipStream_t st) {
    const bool vec = (N % 8 == 0) && N <= 2048 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const size_t lds = vec ? (size_t)max(1, 256 / (N / 8)) * N * sizeof(float) : (N <= 256 ? (size_t)(256 / N) * N * sizeof(float) : 0);
    if (is_bf16) {
        if (vec) hipLaunchKernelGGL((colsum_kernel<bf16, true>), dim3(nblk), dim3(256), lds, st, reinterpret_cast<const bf16*>(x), partial, M, N, nblk);
        else hipLaunchKernelGGL((colsum_kernel<bf16, false>), dim3(nblk), dim3(256), lds, st, reinterpret_cast<const bf16*>(x), partial, M, N, nblk);
    } else {
        if (vec) hipLaunchKernelGGL((colsum_kernel<float, true>), dim3(nblk), dim3(256), lds, st, reinterpret_cast<const float*>(x), partial, M, N, nblk);
        else hipLaunchKernelGGL((colsum_kernel<float, false>), dim3(nblk), dim3(256), lds, st, reinterpret_cast<const float*>(x), partial, M, N, nblk);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + kFinalCols - 1) / kFinalCols), dim3(256), 0, st, partial, out, N, nblk);
    return hipGetLastError();
}

hipError_t cast_launch(const void* x, void* y, int64_t n, int in_bf16, int out_bf16, hipStream_t st) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), blk(256);
    if (!in_bf16 && out_bf16) hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, blk, 0, st, reinterpret_cast<const float*>(x), reinterpret_cast<bf16*>(y), n);
    else if (in_bf16 && !out_bf16) hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, blk, 0, st, reinterpret_cast<const bf16*>(x), reinterpret_cast<float*>(y), n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace hvc
