// Internal host-side argument blocks and launcher prototypes (one launcher per .hip file).
// The public C ABI in include/hvc_hip.h is implemented on top of these in capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvc {

// Run-time switches: kernel-form pins for A/B timing and for the parity tests that must reach every shipped instantiation.
// Atomic ints read at launch (launches also come from torch's autograd worker thread, so a getenv() per launch would race with a
// setenv() on the Python thread); the initial value of each is taken ONCE, when the library is loaded, from the environment
// variable of the same name.  Set through the C ABI: hvc_set_option("HVC_ATTN_FWD_ROWS", 64).
enum Option {
    kOptAttnFwdRows = 0,     // HVC_ATTN_FWD_ROWS     0 = by size, 32 / 64 pin the forward kernel
    kOptAttnFwdWaves,        // HVC_ATTN_FWD_WAVES    0 = by size, 4 / 8 pin the 64-row forward's workgroup form
    kOptAttnBwdWaves,        // HVC_ATTN_BWD_WAVES    0 = by size, 4 / 8 pin the dQ and dK/dV workgroup form
    kOptAttnExtraLds,        // HVC_ATTN_EXTRA_LDS    bytes added to every attention launch's LDS request (occupancy experiments)
    kOptAttnPipe,            // HVC_ATTN_PIPE         1 = software-pipelined attention kernels where their shape conditions hold (default), 0 = the phase-separated twins, 2 = pipelined whenever legal (tests: small grids too)
    kOptGemmPersistent,      // HVC_GEMM_PERSISTENT   1 = persistent token-matrix GEMMs (default), 0 = one tile per workgroup
    kOptGemmStagger,         // HVC_GEMM_STAGGER      start-up stagger of the second workgroup of a CU (>= 0)
    kOptGemmHalfTile,        // HVC_GEMM_HALF_TILE    1 = 64 x 128 tiles on shapes that under-fill the chip (default)
    kOptFp8Mx,               // HVC_FP8_MX            1 = 32x32x64 f8f6f4 forward (round-3 experiment kernel)
    kOptConvForceAddr64,     // HVC_CONV_FORCE_ADDR64 1 = 64-bit addressed gather on every convolution (test hook)
    kOptLossFused,           // HVC_LOSS_FUSED        1 = one-pass SSIM + L1 kernels for the 11-voxel window (default), 0 = three axis passes
    kOptCount
};
int option(Option o);
int cu_count();              // compute units of the current device (cached)

struct AttnArgs {
    const void *q, *k, *v;
    void* o;
    float* lse;          // [B*H][Nq], natural log
    const void* dout;
    float* delta;        // [B*H][Nq] workspace (bwd)
    void *dq, *dk, *dv;
    int B, H, Nq, Nk, D;
    int64_t q_sb, q_sn, q_sh;
    int64_t k_sb, k_sn, k_sh;
    int64_t v_sb, v_sn, v_sh;
    int64_t o_sb, o_sn, o_sh;
    int64_t do_sb, do_sn, do_sh;
    int64_t dq_sb, dq_sn, dq_sh;
    int64_t dk_sb, dk_sn, dk_sh;
    int64_t dv_sb, dv_sn, dv_sh;
    float scale;
    uint32_t seed_lo, seed_hi;
    const uint32_t* seed_ctr;   // device step counter folded into the seed (hvc_set_seed_counter), or null
    uint32_t drop_thresh;   // 0 = no dropout; element dropped when its 16-bit lot < thresh
    float keep_scale;       // 1 / (1 - p)
    int vec;                // 16-byte vector access legal for every operand
    int phases;             // bwd: bit 0 delta, bit 1 dK/dV kernel, bit 2 dQ kernel (0 = all)
    int qsplit;             // dK/dV kernel: number of query-range slices per key block (filled by the launcher)
    float* dkv_partial;     // [2][qsplit][B*H][Nk][D] fp32 partial dK / dV slabs when qsplit > 1
    int64_t partial_floats; // capacity of dkv_partial
    int is_bf16;
};
hipError_t attention_launch(const AttnArgs& a, bool bwd, hipStream_t st);
int attention_bwd_qsplit(int B, int H, int Nq, int Nk);
int64_t attention_fp8_workspace_bytes(int B, int H, int Nk, int D);
hipError_t attention_fp8_launch(const AttnArgs& a, void* workspace, hipStream_t st);      // forward, fp8 (e4m3) MFMA products

enum GemmAct { kActNone = 0, kActGelu = 1, kActGeluGrad = 2 };

// Division by a launch-time constant for operands below 2^31 (Granlund-Montgomery round-up form):
// n / d == (mulhi(n, m) + n) >> l  with l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1.
struct FastDiv { uint32_t d, m, l; };
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.l = 0;
    while ((1ull << f.l) < d) ++f.l;
    f.m = (uint32_t)((((1ull << f.l) - d) << 32) / d + 1);
    return f;
}

// Patch gather of a convolution on channels-last activations, evaluated inside the GEMM operand loader instead of
// being written out by im2col (patch row m = output position, patch column k = tap * C + c, C % 8 == 0).
struct ConvGather {
    const void* src;        // [B][SD][SH][SW][C]
    int C, SD, SH, SW;
    int KD, KH, KW, stride, PD, PH, PW;
    int flip;               // taps run backwards (the stride-1 input gradient gathers dy with the mirrored kernel)
    int64_t M;              // B * OD * OH * OW patch rows
    int K;                  // KD * KH * KW * C patch columns
    uint32_t bytes;         // size of the activation tensor when it is below 4 GiB (buffer-addressed gather), else 0
    FastDiv dC, dKW, dKH, dOW, dOH, dOD;
};

struct GemmArgs {
    const void* A;   // M x K   (k-contiguous: A[i*lda + k];  k-major: A[k*lda + i])
    const void* B;   // N x K   (k-contiguous: B[j*ldb + k];  k-major: B[k*ldb + j])
    void* C;         // M x N row-major, ldc
    int M, N, K;
    int64_t lda, ldb, ldc;
    int a_kmajor, b_kmajor;
    float alpha;
    const float* bias;       // [N] or null
    int act;                 // GemmAct
    void* aux;               // act=gelu: optional pre-activation store; act=gelu_grad: pre-activation input (C dtype, ldc)
    void* zsave;             // optional [M][N] (operand dtype, ldz): value before gate / residual (saved for dgate)
    int64_t ldz;
    const float* gate;       // [M / rows_per_batch][N] or null
    const float* residual;   // [M][N] fp32, ldr, or null
    int64_t ldr;
    int residual_rows;       // > 0: residual has this many rows and is indexed by i % residual_rows (pos_embed broadcast)
    int rows_per_batch;
    uint32_t seed_lo, seed_hi, drop_thresh;
    const uint32_t* seed_ctr;   // device step counter folded into the seed, or null
    float keep_scale;
    int in_bf16, out_bf16;
    int vec_a, vec_b;
    int vec_epi;             // every epilogue operand allows 16-byte row-segment access
    float* workspace;        // optional split-K scratch (fp32), workspace_floats long
    int64_t workspace_floats;
    int splitk;              // filled in by the launcher
    int epi_simple;          // filled in by the launcher: C = alpha A B^T (+ bias), none of the other epilogue options
    int persistent;          // filled in by the launcher: 512 workgroups walk the tile list (see gemm_kernel)
    // optional output row map (one parity class of a strided convolution's input gradient writes its rows into the interleaved
    // positions of dx): row i = ((b * oD + d) * oH + h) * oW + w  ->  element offset o_base + b o_sb + d o_sd + h o_sh + w o_sw
    int omap;
    FastDiv oW, oH, oD;
    int64_t o_base, o_sb, o_sd, o_sh, o_sw;
    int gather;              // 0: both operands in memory; 1: A = conv patches (k-contiguous view), A / lda unused;
                             // 2: B = conv patches (k-major view: contraction over patch rows), B / ldb unused;
                             // 3 / 4: as 1 / 2 for activation tensors of 4 GiB and more (64-bit addressing)
    ConvGather cg;
};
int64_t gemm_workspace_floats(int M, int N, int K);
hipError_t gemm_launch(const GemmArgs& g, hipStream_t st);

struct LnArgs {
    const float* x;          // [rows][C] fp32 (residual stream)
    const float* gamma;      // [C]
    const float* beta;       // [C]
    const float* scale;      // [rows / rows_per_batch][C] or null   (AdaLN: y = ln * (1 + scale) + shift)
    const float* shift;
    void* y;                 // [rows][C] out dtype
    float* mean;             // [rows]
    float* rstd;             // [rows]
    const void* dy;          // bwd
    const float* dres;       // bwd: optional fp32 gradient already flowing on the residual stream (added into dx)
    float* dx;               // bwd: [rows][C] fp32
    float* partial;          // bwd workspace: [nblk][4][C]  (dgamma, dbeta, dscale, dshift partials)
    float* dgamma;           // [C]
    float* dbeta;            // [C]
    float* dscale;           // [nbatch][C] or null
    float* dshift;
    int rows, C, rows_per_batch;
    float eps;
    int out_bf16;
    int blocks_per_batch;    // bwd partial layout
};
hipError_t layernorm_fwd_launch(const LnArgs& a, hipStream_t st);
hipError_t layernorm_bwd_launch(const LnArgs& a, hipStream_t st);
int layernorm_bwd_blocks_per_batch(int rows_per_batch);

struct BranchArgs {
    const float* dy;      // [rows][N] fp32 gradient on the residual stream
    const void* z;        // [rows][N] branch output saved by the forward (needed for dgate) or null
    const float* gate;    // [rows / rows_per_batch][N] or null
    void* dz;             // [rows][N] out: gate * dy in the GEMM dtype
    float* partial;       // workspace [nbatch * blocks_per_batch][2][N]
    float* dgate;         // [nbatch][N] or null
    float* dbias;         // [N] or null
    int rows, N, rows_per_batch, blocks_per_batch;
    int out_bf16;
    uint32_t seed_lo, seed_hi, drop_thresh;   // output dropout applied by the forward GEMM epilogue
    const uint32_t* seed_ctr;   // device step counter folded into the seed, or null
    float keep_scale;
};
hipError_t branch_bwd_launch(const BranchArgs& a, hipStream_t st);
int rowops_blocks(int rows);
hipError_t colsum_launch(const void* x, float* partial, float* out, int M, int N, int nblk, int is_bf16, hipStream_t st);
hipError_t cast_launch(const void* x, void* y, int64_t n, int in_bf16, int out_bf16, hipStream_t st);

struct DrrArgs {
    const void* vol;    // [B][D][H][W]
    void* out;          // axis 0: [B][H][W]; axis 2: [B][D][H] (or [B][H][D] when transpose_out)
    const void* dout;   // bwd
    void* dvol;         // bwd
    int B, D, H, W;
    int axis;           // 0 = project along D, 2 = project along W
    int exp_mode;       // 1: Beer-Lambert exp(-mu (v + 1)), 0: plain intensity
    float mu;
    float out_scale;    // 1 for sum, 1/len for mean
    float clamp_min;    // -inf for none
    int transpose_out;
    int is_bf16;
};
hipError_t drr_fwd_launch(const DrrArgs& a, hipStream_t st);
hipError_t drr_bwd_launch(const DrrArgs& a, hipStream_t st);

struct ConvGeom {
    int B, C;               // batch, source channels
    int SD, SH, SW;         // source extent (channels-last [B][SD][SH][SW][C])
    int OD, OH, OW;         // patch-grid extent
    int KD, KH, KW, stride, PD, PH, PW;
    int64_t M;              // B * OD * OH * OW
    int64_t Kp;             // row pitch of the patch matrix (>= KD*KH*KW*C, multiple of 8)
};
hipError_t im2col_launch(const ConvGeom& g, const void* src, void* col, int is_bf16, hipStream_t st);
hipError_t col2im_launch(const ConvGeom& g, const void* dcol, void* dsrc, int is_bf16, hipStream_t st);
hipError_t trilinear_launch(const float* src, float* dst, int B, int d, int h, int w, int D, int H, int W, bool align_corners, bool bwd,
                            hipStream_t st);
int64_t trilinear_bwd_workspace_floats(int B, int d, int h, int w, int D, int H, int W);
hipError_t trilinear_bwd_separable_launch(const float* dout, float* dsrc, float* workspace, int B, int d, int h, int w, int D, int H, int W,
                                          bool align_corners, hipStream_t st);

// Single-channel convolutions as streaming kernels (conv_direct.hip): Conv3d(1 -> 32 | 64, k3, p1, stride 1 | 2) forward and weight / bias
// gradient, Conv3d(C -> 1, k1) forward and backward.  bf16 operands, channels-last.
struct ConvC1Args {
    const void* x;           // [B][SD][SH][SW] bf16 (one channel)
    const void* w2d;         // [Cout][32] bf16: taps 27 .. 31 are padding
    const float* bias;       // [Cout] or null
    void* y;                 // [B][OD][OH][OW][Cout] bf16
    const void* dy;          // the same shape (weight gradient)
    float* workspace;        // weight gradient: conv_c1_dw_parts() x Cout x 32 floats (one partial per workgroup)
    int B, SD, SH, SW, Cout, stride;
    int OD, OH, OW, tiles_x, tiles_y, tiles_z, ntiles;      // filled by the launchers
};
bool conv_c1_supported(int Cout, int stride);
int conv_c1_dw_parts(int B, int SD, int SH, int SW, int stride);
hipError_t conv_c1_fwd_launch(ConvC1Args a, hipStream_t st);
hipError_t conv_c1_dw_launch(ConvC1Args a, float* dw, hipStream_t st);      // dw: [Cout][32] fp32, column 27 = bias gradient
hipError_t conv_c1_dx_launch(ConvC1Args a, hipStream_t st);                // stride 1: dy in a.dy, w^T [32][Cout] in a.w2d, dx in a.y
// Conv3d(CI -> CO, k3, s1, p1), CI, CO in {32, 64}, bf16 channels-last, through an LDS halo tile (conv_direct.hip).
struct Conv3Args {
    const void* x;           // [B][D][H][W][CI]
    const void* wfrag;       // weight fragments [27 taps][CI/16][CO/32][64 lanes][8]: element j of lane l = W[32 nt + (l & 31)][tap][16 ck + 8 (l >> 5) + j]
    const float* bias;       // [CO] or null
    void* y;                 // [B][D][H][W][CO]
    int B, D, H, W, CI, CO;
    int tiles_x, tiles_y, tiles_z;      // filled by the launcher
};
bool conv3_halo_supported(int CI, int CO);
hipError_t conv3_halo_launch(Conv3Args a, hipStream_t st);
bool conv_o1_supported(int C);
int conv_o1_bwd_blocks(int64_t M, int C);
hipError_t conv_o1_fwd_launch(const void* x, const void* w, const float* bias, void* y, int64_t M, int C, hipStream_t st);
// dwb: [C + 1] fp32 = dW, then the bias gradient; workspace: conv_o1_bwd_blocks() x (C + 1) floats; dx may be null
hipError_t conv_o1_bwd_launch(const void* x, const void* dy, const void* w, void* dx, float* dwb, float* workspace, int64_t M, int C, hipStream_t st);

struct PoolGeom { int N, H, W, C, HP, WP, k, s, p; };
struct NormArgs {
    const void* x; void* y; const void* dy; void* dx;
    const float* gamma; const float* beta;
    float* stats;            // GN: [B][G][2] (mean, rstd); BN: [C][2]
    float* partial;          // [B][chunks][2][C]
    float* gsum;             // GN bwd: [B][G][2]; BN bwd: [C][2] means of ds, ds*xhat
    float* dgamma; float* dbeta;
    float* running_mean; float* running_var;
    uint8_t* amax;
    int B, P, C, G;
    float eps, momentum;
    int training, is_bf16;
    int act;                 // GroupNorm epilogue activation: 0 SiLU, 1 GELU(erf)
};
int norm_chunks(int P);
hipError_t groupnorm_silu_fwd_launch(const NormArgs& a, hipStream_t st);
hipError_t groupnorm_silu_bwd_launch(const NormArgs& a, hipStream_t st);
hipError_t bn_relu_pool_fwd_launch(const NormArgs& a, const PoolGeom& pg, hipStream_t st);
hipError_t bn_relu_pool_bwd_launch(const NormArgs& a, const PoolGeom& pg, hipStream_t st);

struct LossArgs {
    const float* pred; const float* target;
    float* out;              // [3] total, l1, ssim-loss
    float* gmaps;            // [3][nvox] derivative maps (fwd out / bwd in)
    float* workspace;
    const float* gscale;     // device [3]: upstream gradients of (total, l1, ssim_loss), or null (= 1, 0, 0)
    float* dpred;
    int B, D, H, W, window;
    float l1_w, ssim_w;
};
int loss_blocks(int64_t nvox);
hipError_t ssim_l1_fwd_launch(const LossArgs& a, hipStream_t st);
hipError_t ssim_l1_bwd_launch(const LossArgs& a, hipStream_t st);

struct TvArgs {
    const float* vol;        // (B, D, H, W) fp32
    float* out;              // [3] mean sqrt(diff^2 + eps) along D, H, W
    float* workspace;        // 3 * loss_blocks(nvox)
    const float* gscale;     // device [3]: upstream gradients of the three means
    float* dvol;
    int B, D, H, W;
    float eps;
};
hipError_t tv3d_fwd_launch(const TvArgs& a, hipStream_t st);
hipError_t tv3d_bwd_launch(const TvArgs& a, hipStream_t st);

struct SpecArgs {
    const float* pred; const float* target;   // complex spectra, interleaved (re, im): [B][D][H][W][2]
    float* out;              // [2] (1/N) sum | |P| - |T| | over the low- / high-frequency cells
    float* workspace;        // 2 * loss_blocks(B*D*H*W)
    const float* gscale;     // device [2]: upstream gradients of the two terms
    float* dpred;            // [B][D][H][W][2]
    int B, D, H, W;
};
hipError_t spec_l1_fwd_launch(const SpecArgs& a, hipStream_t st);
hipError_t spec_l1_bwd_launch(const SpecArgs& a, hipStream_t st);

struct ResizeLossArgs {
    const float* proj;       // (B, h, w) fp32 contiguous
    const float* target;     // (B, S1, S2) fp32, rows contiguous, batch stride target_bstride elements
    float* out;              // [1] mean |r - t| (mode 0) or mean (r - t)^2 (mode 1)
    float* workspace;        // resize_loss_blocks(B, S1, S2) floats
    const float* gscale;     // device [1]: upstream gradient of the scalar
    float* dres;             // (B, S1, S2): d loss / d resized image (input of the resize adjoint)
    int B, h, w, S1, S2;
    int64_t target_bstride;
    int align_corners, mode;
};
int resize_loss_blocks(int B, int S1, int S2);
hipError_t resize_loss_fwd_launch(const ResizeLossArgs& a, hipStream_t st);
hipError_t resize_loss_grad_launch(const ResizeLossArgs& a, hipStream_t st);

struct ViewGapArgs {
    const void* f;           // (B*V, P, E) channels-last feature maps, fp32 or bf16
    float* mean;             // (B, P, E) fp32: mean over the V views
    float* pooled;           // (B, E) fp32: mean over P of `mean`
    float* workspace;        // B * view_gap_chunks(P) * E floats
    const float* dmean;      // bwd: (B, P, E) or null
    const float* dpooled;    // bwd: (B, E) or null
    void* df;                // bwd: (B*V, P, E) in the dtype of f
    int B, V, P, E;
    int is_bf16;
};
int view_gap_chunks(int P);
hipError_t view_mean_gap_fwd_launch(const ViewGapArgs& a, hipStream_t st);
hipError_t view_mean_gap_bwd_launch(const ViewGapArgs& a, hipStream_t st);

}  // namespace hvc
