/*
 * hvc_hip.h -- C ABI of libhvc_hip.so: the MI355X (gfx950) native compute path behind the
 * Hybrid-ViT-Cascade Python class surface.
 *
 * The reference (kanadm12/Hybrid-ViT-Cascade) has NO native / FFI layer: every entry point below
 * replaces a sequence of stock ATen ops issued by the reference's Python modules.  Each declaration
 * cites the reference lines whose arithmetic it replaces.  The host-side mirror of the reference's
 * module surface (hybrid-vit-cascade_amd/models/...) binds these symbols with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless stated; no torch types cross this boundary
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is asynchronous
 *     on that stream, allocates nothing and keeps no global state, so calls can be captured in
 *     a hipGraph
 *   - dtype codes: HVC_F32 = 0, HVC_BF16 = 1.  MFMA kernels compute HVC_F32 operands as split
 *     bf16 (hi + lo) products with fp32 accumulation (~1e-5 relative), HVC_BF16 operands natively
 *   - return value: 0 on success, a negative HVC_E_* code on a bad argument, a positive hipError_t
 *     on a launch failure; hvc_last_error() returns a static message for the calling thread
 *   - tensors are dense row-major with the strides stated per call (in ELEMENTS)
 */
#ifndef HVC_HIP_H
#define HVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVC_F32 0
#define HVC_BF16 1

#define HVC_E_BADARG (-1)
#define HVC_E_UNSUPPORTED (-2)

#define HVC_ABI_VERSION 1

/* Library / device probes (host side only, no kernel launch). */
int hvc_abi_version(void);
const char* hvc_last_error(void);
/* Device-resident dropout step counter (for hipGraph capture of a training step: a replayed graph passes the same seed
 * arguments every time, so the per-step variation of every dropout mask must live in device memory).  After
 * hvc_set_seed_counter(ptr) every kernel that draws dropout masks (attention probabilities, GEMM-epilogue dropouts and their
 * backward passes) offsets its 64-bit seed argument by the 32-bit word at ptr, read on the device at kernel start; forward and
 * backward of one step see the same value and so regenerate the same masks.  hvc_seed_counter_advance enqueues a one-thread
 * kernel adding `step` to the word - captured as the first node of the step's graph.  NULL restores plain seeds.  The
 * setting is process-wide (one process per GPU): autograd runs backward passes on its own worker thread, which must see the
 * counter the forward saw.  (The reference draws its masks from torch's Philox stream: train_direct_4gpu.py:65-71.) */
int hvc_set_seed_counter(const uint32_t* device_counter);
int hvc_seed_counter_advance(uint32_t* device_counter, uint32_t step, void* stream);
/* Detaches the counter only if the library still points at `device_counter` (compare-and-swap): the finaliser of an older
 * graphed step must not detach the counter a newer one has installed.  Returns 0 either way. */
int hvc_clear_seed_counter_if(const uint32_t* device_counter);

/* Run-time switches of the library (kernel-form pins for A/B timing and for parity tests that must reach every shipped
 * instantiation of a kernel; no reference counterpart - the reference has one code path per op).  Names are those of the
 * environment variables that give the initial values when the library is loaded: HVC_ATTN_FWD_ROWS (0 | 32 | 64),
 * HVC_ATTN_FWD_WAVES (0 | 4 | 8), HVC_ATTN_BWD_WAVES (0 | 4 | 8), HVC_ATTN_PIPE (1 | 0 | 2), HVC_ATTN_EXTRA_LDS (bytes),
 * HVC_GEMM_PERSISTENT (1 | 0), HVC_GEMM_STAGGER (>= 0), HVC_GEMM_HALF_TILE (1 | 0), HVC_FP8_MX (0 | 1),
 * HVC_CONV_FORCE_ADDR64 (0 | 1), HVC_LOSS_FUSED (1 | 0).  Values are atomic ints read at launch time, safe to set from any thread between launches;
 * unknown names and negative values return HVC_E_BADARG. */
int hvc_set_option(const char* name, int value);
int hvc_get_option(const char* name, int* value);

/* Fills CU count and wavefront size of the current device and its gcnArchName; 0 on success. */
int hvc_device_info(int* cu_count, int* wavefront, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * Fused attention core.  Replaces models/vit_components.py:41-51 (MultiHeadSelfAttention.forward:
 * reshape/permute of qkv, q@k^T*scale, softmax, attn_drop, @v, transpose/reshape) and
 * models/vit_components.py:95-113 (MultiHeadCrossAttention.forward, same core with Nq != Nk).
 *
 * x[b][n][h][d] lives at  ptr[b*sb + n*sn + h*sh + d]  (head dim contiguous), so q/k/v can alias
 * the packed projection output and `o` is written directly in the (B, N, h*d) layout.
 * D must be 32 or 64.  lse is [B*H][Nq] fp32 (natural log of the softmax denominator).
 * p_drop > 0 applies the attention-probability dropout of attn_drop with a counter-based RNG keyed
 * by (seed, b, h, q, k): the backward call with the same seed regenerates the same mask.
 * ---------------------------------------------------------------------------------------------- */
int hvc_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                      int B, int H, int Nq, int Nk, int D,
                      int64_t q_sb, int64_t q_sn, int64_t q_sh,
                      int64_t k_sb, int64_t k_sn, int64_t k_sh,
                      int64_t v_sb, int64_t v_sn, int64_t v_sh,
                      int64_t o_sb, int64_t o_sn, int64_t o_sh,
                      float scale, float p_drop, uint64_t seed, int dtype, void* stream);

/* Forward with fp8 (OCP e4m3) MFMA products - v_mfma_f32_32x32x16_fp8_fp8 for Q K^T and P V - the "fp8 MFMA attention" of
 * BASELINE configs[4] (cascade stage 3: direct_regression/progressive_cascade/model_progressive.py:219-316 runs these
 * attention modules at 8 heads x 32).  Same contract as hvc_attention_fwd for bf16 operands with 16-byte addressable rows;
 * softmax statistics, dropout lots and the outputs (bf16 o, fp32 lse) are those of the bf16 kernel, K / V / P are rounded to
 * e4m3 (3 mantissa bits: stated tolerance 6e-2 relative Frobenius error on o for white-noise operands, 5e-2 on structured ones).  The backward pass is hvc_attention_bwd (bf16)
 * with this forward's o / lse.  workspace: hvc_attention_fwd_fp8_workspace BYTES (fp8 images of K and of V transposed). */
int64_t hvc_attention_fwd_fp8_workspace(int B, int H, int Nk, int D);
int hvc_attention_fwd_fp8(const void* q, const void* k, const void* v, void* o, float* lse, void* workspace,
                          int B, int H, int Nq, int Nk, int D,
                          int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                          int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                          float scale, float p_drop, uint64_t seed, void* stream);

/* Floats of scratch hvc_attention_bwd needs: B*H*Nq for delta, plus fp32 partial dK/dV slabs when few key blocks
 * (cross-attention) make the dK/dV kernel slice the query range over extra workgroups. */
int64_t hvc_attention_bwd_workspace(int B, int H, int Nq, int Nk, int D);

/* Gradient of the above (autograd of the same reference lines).  workspace: hvc_attention_bwd_workspace floats.
 * dq/dk/dv use the strides of q/k/v respectively; dout uses the strides of o.
 * Three launches: delta = rowsum(dO * O) (bit 0 of `phases`), the dK/dV kernel (bit 1), the dQ kernel (bit 2);
 * phases = 0 or 7 runs all of them, a sub-mask lets a caller time them separately (they must still be issued in
 * that order on one stream). */
int hvc_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                      const float* lse, float* workspace, void* dq, void* dk, void* dv,
                      int B, int H, int Nq, int Nk, int D,
                      int64_t q_sb, int64_t q_sn, int64_t q_sh,
                      int64_t k_sb, int64_t k_sn, int64_t k_sh,
                      int64_t v_sb, int64_t v_sn, int64_t v_sh,
                      int64_t o_sb, int64_t o_sn, int64_t o_sh,
                      float scale, float p_drop, uint64_t seed, int phases, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   z = dropout(act(alpha * A B^T + bias));  C = residual + gate_b * z
 *   C[i][j] = sum_k A(i,k) B(j,k);  an operand is k-contiguous (X[i*ld + k]) or, with *_kmajor = 1,
 *   k-major (X[k*ld + i]) -- so dx = dy W and dW = dy^T x read W / activations in place.
 * Replaces nn.Linear forward/backward at models/vit_components.py:26,28,41,54,74,75,77,95,98,115,
 * 131,144 and models/hybrid_vit_backbone.py:75-81 (Linear, GELU(erf), Dropout, Linear, Dropout),
 * plus the gated residual adds at models/hybrid_vit_backbone.py:123,128,139.
 *   act: 0 none | 1 GELU(erf) (aux != NULL additionally receives the pre-activation, dtype/ld of C)
 *        | 2 multiply by GELU'(aux)  (backward of 1; aux = saved pre-activation)
 *   zsave: optional [M][N] copy of z in in_dtype with leading dimension ldz (kept for the gate gradient);
 *   bias: [N] fp32 or NULL;  gate: [M / rows_per_batch][N] fp32 or NULL;
 *   residual: [M][N] fp32 with leading dimension ldr, or NULL; with residual_rows > 0 it has that many
 *   rows and row i adds residual[i % residual_rows] (the pos_embed add of
 *   models/hybrid_vit_backbone.py:258 fused into the last stem convolution).
 *   in_dtype / out_dtype: (BF16,BF16), (BF16,F32) or (F32,F32).
 * ---------------------------------------------------------------------------------------------- */
int hvc_gemm(const void* A, const void* B, void* C, int M, int N, int K,
             int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor, float alpha,
             const float* bias, int act, void* aux, void* zsave, int64_t ldz,
             const float* gate, const float* residual, int64_t ldr, int residual_rows, int rows_per_batch,
             float p_drop, uint64_t seed, float* workspace, int64_t workspace_floats,
             int in_dtype, int out_dtype, void* stream);
/* Floats of split-K scratch worth passing as `workspace` for this shape (0 = none needed).  With a
 * workspace and a plain epilogue (alpha only) a small-output / deep-contraction product -- every
 * weight gradient dW = dy^T x -- is sliced over workgroups along K into fp32 slabs that a second
 * kernel sums in a fixed order (bitwise reproducible).  workspace may be NULL. */
int64_t hvc_gemm_workspace(int M, int N, int K);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (+ optional AdaLN modulate y = ln(x) * (1 + scale_b) + shift_b).
 * Replaces nn.LayerNorm at models/hybrid_vit_backbone.py:84-86,229 and the modulate lines
 * models/hybrid_vit_backbone.py:120-121,136-137.  x: [rows][C] fp32, y: [rows][C] out_dtype,
 * gamma/beta: [C] fp32, scale/shift: [rows / rows_per_batch][C] fp32 or both NULL. C <= 1024.
 * mean / rstd: [rows] fp32 saved for the backward.
 * ---------------------------------------------------------------------------------------------- */
int hvc_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                      const float* scale, const float* shift, void* y, float* mean, float* rstd,
                      int rows, int C, int rows_per_batch, float eps, int out_dtype, void* stream);

/* Number of floats of scratch hvc_layernorm_bwd needs. */
int64_t hvc_layernorm_bwd_workspace(int rows, int C, int rows_per_batch);

/* dx = dres (optional, fp32, already on the residual stream) + LN backward;  dgamma/dbeta: [C];
 * dscale/dshift: [rows / rows_per_batch][C] (required iff scale != NULL). */
int hvc_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* beta,
                      const float* scale, const float* mean, const float* rstd, const float* dres,
                      float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift,
                      float* workspace, int rows, int C, int rows_per_batch, int dy_dtype,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of a (gated) residual branch  x_out = x + gate_b * z
 * (models/hybrid_vit_backbone.py:123,128,139) where z = dropout(Linear(..)) came out of hvc_gemm
 * with the same (p_drop, seed):  dz = gate_b * dy * dropmask / (1 - p) (cast to out_dtype),
 * dgate_b[n] = sum_rows dy*z,  dbias[n] = sum_rows dz.  dy: [rows][N] fp32; z: [rows][N] out_dtype
 * or NULL (then dgate must be NULL); gate NULL = ungated (cross-attention branch).
 * ---------------------------------------------------------------------------------------------- */
int64_t hvc_branch_bwd_workspace(int rows, int N, int rows_per_batch);
int hvc_branch_bwd(const float* dy, const void* z, const float* gate, void* dz,
                   float* dgate, float* dbias, float* workspace,
                   int rows, int N, int rows_per_batch, float p_drop, uint64_t seed,
                   int out_dtype, void* stream);

/* Column sum of x[M][N] (bias gradient of a Linear, models/hybrid_vit_backbone.py:76).
 * workspace: hvc_colsum_workspace(M, N) floats. */
int64_t hvc_colsum_workspace(int M, int N);
int hvc_colsum(const void* x, float* out, float* workspace, int M, int N, int dtype, void* stream);

/* Elementwise dtype cast (fp32 <-> bf16) of n contiguous elements: parameter / activation casts
 * that torch.autocast performs implicitly in the reference trainers
 * (direct_regression/train_direct_4gpu.py:65). */
int hvc_cast(const void* x, void* y, int64_t n, int in_dtype, int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DRR ray-sum projection.  vol: [B][D][H][W].
 *   axis 0: out[b][h][w] = max(clamp_min, out_scale * sum_d f(vol))
 *   axis 2: out[b][d][h] (or out[b][h][d] when transpose_out) = max(clamp_min, out_scale * sum_w f(vol))
 *   f(v) = exp(-mu (v + 1)) when exp_mode else v.  Pass clamp_min = -INFINITY for no clamp.
 * Replaces models/diagnostic_losses.py:45-63 (DRRRenderer.forward: exp_mode=1, mu=0.3, out_scale=1,
 * clamp_min=1e-6, transpose_out=1 for angle 90) and
 * direct_regression/progressive_cascade/loss_multiscale.py:260-267 (generate_drr: exp_mode=0,
 * out_scale=1/D or 1/W, no clamp, no transpose).
 * ---------------------------------------------------------------------------------------------- */
int hvc_drr_fwd(const void* vol, void* out, int B, int D, int H, int W, int axis, int exp_mode,
                float mu, float out_scale, float clamp_min, int transpose_out, int dtype,
                void* stream);
int hvc_drr_bwd(const void* vol, const void* out, const void* dout, void* dvol,
                int B, int D, int H, int W, int axis, int exp_mode, float mu, float out_scale,
                float clamp_min, int transpose_out, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution stems as im2col + hvc_gemm on CHANNELS-LAST activations [B][D][H][W][C] (2-D: D = 1).
 * Replaces nn.Conv3d(k3, s1|s2, p1) at models/hybrid_vit_backbone.py:195-210 and
 * nn.Conv2d(k7 s2 p3 / k3 s1 p1) at models/diagnostic_losses.py:82,87,92, forward and backward:
 *   y  = hvc_gemm(col, W2d[Cout][taps*C], bias)             col = hvc_im2col(x)
 *   dx = hvc_col2im(hvc_gemm(dy, W2d, b_kmajor))            (gather form, deterministic)
 *   dW = hvc_gemm(dy, col, a_kmajor, b_kmajor)              (split-K)
 * col is [B*OD*OH*OW][Kp], column index = tap*C + c, tap = (kd*KH + kh)*KW + kw; Kp = taps*C when
 * C % 8 == 0, else taps*C rounded up to a multiple of 8 (zero filled).  O = (S + 2P - K)/stride + 1.
 * Depth slabs: a large volume is convolved in slabs along D; for a slab the caller passes the slab extent as SD,
 * the FRONT padding as PD (the back padding is implied: rows beyond SD read as zero) and the slab's output
 * depth as OD (> 0; 0 = derive from the symmetric formula).
 * ---------------------------------------------------------------------------------------------- */
int hvc_im2col(const void* src, void* col, int B, int C, int SD, int SH, int SW, int KD, int KH, int KW,
               int stride, int PD, int PH, int PW, int OD, int64_t Kp, int dtype, void* stream);
int hvc_col2im(const void* dcol, void* dsrc, int B, int C, int SD, int SH, int SW, int KD, int KH, int KW,
               int stride, int PD, int PH, int PW, int OD, int64_t Kp, int dtype, void* stream);

/* Single-channel convolutions as streaming kernels (bf16, channels-last; csrc/conv_direct.hip) - no patch matrix on either side:
 *   hvc_conv_c1_*  nn.Conv3d(1, Cout, 3, stride, padding=1), Cout = 32 | 64, stride 1 | 2:
 *                  direct_regression/progressive_cascade/model_progressive.py:171,240,260 (cascade glue, 128^3 / 256^3) and the first
 *                  voxel-embed layer models/hybrid_vit_backbone.py:199.  x [B][SD][SH][SW], w2d [Cout][32] (tap = (kd*3 + kh)*3 + kw, columns
 *                  27..31 ignored), y / dy [B][OD][OH][OW][Cout], O = (S - 1)/stride + 1.  hvc_conv_c1_dw returns dw [Cout][32] fp32 whose
 *                  column 27 is the BIAS gradient (sum of dy over positions); workspace = hvc_conv_c1_dw_workspace(...) floats; partial sums
 *                  are added in a fixed order (bitwise reproducible).  Input gradient: hvc_conv_c1_dx below (stride 1).
 *   hvc_conv_o1_*  nn.Conv3d(C, 1, 1) (model_progressive.py:266), C = 8 | 16 | 32 | 64 | 128, on the [M][C] view of the activations:
 *                  y[m] = bias + x[m] . w;  backward: dx[m][c] = dy[m] w[c] (dx may be NULL), dwb[0..C) = sum_m dy[m] x[m][c], dwb[C] = sum_m dy[m];
 *                  workspace = hvc_conv_o1_bwd_workspace(M, C) floats. */
int hvc_conv_c1_fwd(const void* x, const void* w2d, const float* bias, void* y, int B, int SD, int SH, int SW, int Cout, int stride,
                    void* stream);
int64_t hvc_conv_c1_dw_workspace(int B, int SD, int SH, int SW, int Cout, int stride);
int hvc_conv_c1_dw(const void* x, const void* dy, float* dw, float* workspace, int B, int SD, int SH, int SW, int Cout, int stride,
                   void* stream);
/* nn.Conv3d(CI, CO, 3, padding=1), CI, CO in {32, 64}, bf16 channels-last, through an LDS halo tile instead of a 27-fold gather from L2:
 * direct_regression/progressive_cascade/model_progressive.py:263 (detail_enhancer 64 -> 32 at 256^3) and, with mirrored taps and transposed
 * weights, its input gradient.  wfrag: the weights pre-arranged in MFMA fragment order, [27 taps][CI/16][CO/32][64 lanes][8] bf16 with
 * element j of lane l = W[32 nt + (l & 31)][tap][16 ck + 8 (l >> 5) + j], W = (CO, tap = (kd*3 + kh)*3 + kw, CI). */
int hvc_conv3_halo(const void* x, const void* wfrag, const float* bias, void* y, int B, int D, int H, int W, int CI, int CO, void* stream);
/* Input gradient of hvc_conv_c1_fwd at stride 1: dy [B][D][H][W][Cout] bf16, wt = the weights TRANSPOSED, [32 taps][Cout] bf16 with rows 27..31
 * zero, dx [B][D][H][W] bf16.  (Stride 2 - the direct model's first voxel-embed layer, whose input needs no gradient - stays on gemm + col2im.) */
int hvc_conv_c1_dx(const void* dy, const void* wt, void* dx, int B, int D, int H, int W, int Cout, void* stream);
int hvc_conv_o1_fwd(const void* x, const void* w, const float* bias, void* y, int64_t M, int C, void* stream);
int64_t hvc_conv_o1_bwd_workspace(int64_t M, int C);
int hvc_conv_o1_bwd(const void* x, const void* dy, const void* w, void* dx, float* dwb, float* workspace, int64_t M, int C, void* stream);

/* Convolution as an IMPLICIT GEMM (C % 8 == 0): the patch matrix above is never written; the GEMM's operand loader
 * gathers the 16-byte channel vectors of each tap straight from the channels-last activations (zero outside the volume).
 * Replaces the same nn.Conv3d / nn.Conv2d layers (models/hybrid_vit_backbone.py:195-210, models/diagnostic_losses.py:87,92,
 * direct_regression/progressive_cascade/model_progressive.py:46-49,121-123) without the HBM round trip of col / dcol:
 *   mode 0: out[M][N]  = patches(src)[M][K] . other[N][K]^T (+ bias[N], + residual rows as in hvc_gemm)
 *           forward:  src = x, other = W2d[Cout][taps*C], N = Cout
 *           stride-1 input gradient: src = dy (C = Cout), other = W^T[Cin][taps*Cout] with the taps mirrored
 *           (flip = 1: tap (kd,kh,kw) of the gather is (KD-1-kd, ...) of the column index), pads K-1-P, N = Cin
 *   mode 1: out[N][K]  = other[M][N]^T . patches(src)[M][K]   (weight gradient, fp32, deterministic split-K;
 *           other = dy[M][Cout], N = Cout)
 * M = B*OD*OH*OW, K = taps*C, O = (S + 2P - K)/stride + 1; M, K < 2^31. */
int hvc_conv_gemm(int mode, const void* src, const void* other, void* out, int B, int C, int SD, int SH, int SW,
                  int KD, int KH, int KW, int stride, int PD, int PH, int PW, int flip, int N, int64_t ld_other,
                  int64_t ld_out, const float* bias, const float* residual, int64_t ldr, int residual_rows,
                  float* workspace, int64_t workspace_floats, int in_dtype, int out_dtype, void* stream);

/* Input gradient of a STRIDED convolution (stride s > 1) without the dcol matrix and col2im: the input positions split into
 * s^3 parity classes (position = s * i' + class per axis); within a class every position is reached by the same taps, so
 * its gradient is a stride-1 implicit GEMM over dy with a (1..ceil(K/s))^3 window, written straight to the class's interleaved
 * positions of dx.  One call per class; a class no tap reaches (K < s) is rejected - those dx values are zero.
 *   wclass: the class's columns of W^T[Cin][taps * Cout] re-ordered class-major: classes in (cd, ch, cw) lexicographic order,
 *           inside a class taps (td, th, tw) lexicographic with t <-> kernel index k0 - t * s per axis (k0 the largest index of
 *           the class: ascending source offset); hvc_conv_dx_class_columns gives a class's first tap and tap count.
 *           ld_w = row pitch of that matrix (KD*KH*KW*Cout when all classes share one allocation).
 * Replaces the backward of the stride-2 nn.Conv3d / nn.Conv2d layers of models/hybrid_vit_backbone.py:195-210 and
 * direct_regression/progressive_cascade/model_progressive.py:46-49.  Cin % 8 == Cout % 8 == 0. */
int hvc_conv_dx_class_columns(int KD, int KH, int KW, int stride, int PD, int PH, int PW, int cd, int ch, int cw,
                              int* first_tap, int* ntaps);
int hvc_conv_dx_class(const void* dy, const void* wclass, void* dx, int B, int Cout, int OD, int OH, int OW,
                      int Cin, int SD, int SH, int SW, int KD, int KH, int KW, int stride, int PD, int PH, int PW,
                      int cd, int ch, int cw, int64_t ld_w, int dtype, void* stream);

/* Trilinear resize of single-channel fp32 volumes [B][d][h][w] -> [B][D][H][W] and its adjoint:
 * align_corners=1 for F.interpolate at models/hybrid_vit_backbone.py:272; align_corners=0 for the cascade's
 * nn.Upsample / F.interpolate (direct_regression/progressive_cascade/model_progressive.py:170,211,239,294). */
int hvc_trilinear_fwd(const float* src, float* dst, int B, int d, int h, int w, int D, int H, int W,
                      int align_corners, void* stream);
/* Adjoint.  With a workspace of hvc_trilinear_bwd_workspace() floats the three axes are reduced in separable passes
 * (W, then H, then D: 10x less work than gathering the 3-D support per coarse voxel); workspace = NULL runs the
 * single-pass gather. */
int64_t hvc_trilinear_bwd_workspace(int B, int d, int h, int w, int D, int H, int W);
int hvc_trilinear_bwd(const float* dout, float* dsrc, float* workspace, int B, int d, int h, int w, int D, int H, int W,
                      int align_corners, void* stream);

/* Floats of scratch for the four normalisation calls below. */
int64_t hvc_norm_workspace(int B, int P, int C, int G);

/* GroupNorm(G, C) + activation on channels-last x[B][P][C]: act 0 = SiLU (nn.GroupNorm + nn.SiLU at
 * models/hybrid_vit_backbone.py:199-200), act 1 = GELU(erf) (cascade glue, model_progressive.py:39-40,
 * 172-173).  stats: [B][G][2] (mean, rstd), written by fwd, read by bwd.  C in {8,16,32,64,128,256,512}. */
int hvc_groupnorm_act_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats,
                          float* workspace, int B, int P, int C, int G, float eps, int act, int dtype,
                          void* stream);
int hvc_groupnorm_act_bwd(const void* x, const void* dy, void* dx, const float* gamma, const float* beta,
                          const float* stats, float* dgamma, float* dbeta, float* workspace,
                          int B, int P, int C, int G, int act, int dtype, void* stream);

/* BatchNorm2d + ReLU + MaxPool2d(k, s, p) on channels-last x[N][H][W][C] (models/diagnostic_losses.py:83-85,
 * 88-90, 93-94; k = 1 for the last, pool-less stage).  training: batch statistics, running_mean/var
 * updated in place with `momentum` (unbiased variance), else running statistics.  y: [N][HP][WP][C];
 * amax: [N][HP][WP][C] bytes (window-local arg-max, needed when k > 1); stats: [C][2].
 * workspace: hvc_norm_workspace(N, H*W, C, C) floats. */
int hvc_bn_relu_pool_fwd(const void* x, void* y, uint8_t* amax, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* stats, float* workspace,
                         int N, int H, int W, int C, int k, int s, int p, int training, float eps,
                         float momentum, int dtype, void* stream);
int hvc_bn_relu_pool_bwd(const void* x, const void* dy, const uint8_t* amax, void* dx, const float* gamma,
                         const float* beta, const float* stats, float* dgamma, float* dbeta,
                         float* workspace, int N, int H, int W, int C, int k, int s, int p, int training,
                         int dtype, void* stream);

/* Fused training loss  l1_w * mean|p - t| + ssim_w * (1 - mean SSIM_window^3(p, t))  on fp32 volumes
 * [B][D][H][W] (direct_regression/model_direct.py:88-131; window 11, zero padded, divisor window^3).
 * fwd: out3 = (total, l1, ssim_loss); gmaps [3][B*D*H*W] is kept for bwd;
 *      workspace: hvc_ssim_l1_workspace floats.  bwd: dpred = sum_i gscale[i] * d out3[i] / dpred with
 *      gscale a device [3] vector (NULL = (1,0,0)); workspace: 6 * B*D*H*W floats. */
int64_t hvc_ssim_l1_workspace(int B, int D, int H, int W);
int hvc_ssim_l1_fwd(const float* pred, const float* target, float* out3, float* gmaps, float* workspace,
                    int B, int D, int H, int W, int window, float l1_w, float ssim_w, void* stream);
int hvc_ssim_l1_bwd(const float* pred, const float* target, const float* gmaps, const float* gscale,
                    float* dpred, float* workspace, int B, int D, int H, int W, int window, float l1_w,
                    float ssim_w, void* stream);

/* Total variation of (B, 1, D, H, W) fp32 volumes - the per-axis terms of TotalVariationLoss
 * (reference direct_regression/progressive_cascade/loss_multiscale.py:140-188):
 * means3[a] = mean sqrt((v[i+1] - v[i])^2 + eps) over the forward differences along a = D, H, W.  The reference's
 * (sum / 3), clamp(0, 100) and optional |tv_pred - tv_target| are scalar arithmetic on means3 and stay with the caller.
 * workspace: hvc_tv3d_workspace floats.  bwd: dvol = sum_a gscale[a] * d means3[a] / d vol, gscale a device [3] vector. */
int64_t hvc_tv3d_workspace(int B, int D, int H, int W);
int hvc_tv3d_fwd(const float* vol, float* means3, float* workspace, int B, int D, int H, int W, float eps, void* stream);
int hvc_tv3d_bwd(const float* vol, const float* gscale, float* dvol, int B, int D, int H, int W, float eps, void* stream);

/* Magnitude-spectrum L1 terms of FrequencyLoss (reference direct_regression/progressive_cascade/loss_multiscale.py:191-236).
 * pred_spec / target_spec: the 3-D FFTs (over D, H, W; UNSHIFTED, as torch.fft.fftn / rocFFT leave them) of the two volumes as
 * interleaved (re, im) fp32, [B][D][H][W][2].  A cell is "high frequency" when its index is further than min(D,H,W)/4 from
 * (D/2, H/2, W/2) -- the reference's mask as written (:218-231).  out2 = (1/N) sum | |P| - |T| | over the low / the high cells,
 * N = B*D*H*W; the reference's  low + high_freq_weight * high  is scalar arithmetic and stays with the caller.
 * bwd: dpred_spec = sum_i gscale[i] * d out2[i] / d pred_spec (as a real pair per cell), gscale a device [2] vector.
 * workspace: hvc_spectral_l1_workspace floats.  The transform itself is the vendor FFT (rocFFT), called by the host side. */
int64_t hvc_spectral_l1_workspace(int B, int D, int H, int W);
int hvc_spectral_l1_fwd(const float* pred_spec, const float* target_spec, float* out2, float* workspace, int B, int D, int H, int W,
                        void* stream);
int hvc_spectral_l1_bwd(const float* pred_spec, const float* target_spec, const float* gscale, float* dpred_spec,
                        int B, int D, int H, int W, void* stream);

/* Projection-loss epilogue: bilinear resize of a projection (B, h, w) fp32 to (S1, S2) fused with the reduction against
 * the target X-ray (B, S1, S2) (rows contiguous, batch stride target_bstride elements, so a view xrays[:, v] is passed in place):
 *   mode 0: mean |r - t|     - DRRReprojectionLoss, reference direct_regression/progressive_cascade/loss_multiscale.py:269-293
 *                              (align_corners = 0)
 *   mode 1: mean (r - t)^2   - ProjectionLoss, reference models/diagnostic_losses.py:161-169 (align_corners = 1)
 * fwd writes the scalar out1; the resized image is never materialised.  grad writes dresized = gscale[0] * d out1 / d r
 * (B, S1, S2); the caller feeds it to hvc_trilinear_bwd with depth 1 (the adjoint of the resize).  workspace:
 * hvc_resize_loss_workspace floats. */
int64_t hvc_resize_loss_workspace(int B, int S1, int S2);
int hvc_resize_loss_fwd(const float* proj, const float* target, float* out1, float* workspace, int B, int h, int w, int S1, int S2,
                        int64_t target_bstride, int align_corners, int mode, void* stream);
int hvc_resize_loss_grad(const float* proj, const float* target, const float* gscale, float* dresized, int B, int h, int w,
                         int S1, int S2, int64_t target_bstride, int align_corners, int mode, void* stream);

/* Mean over the V views of the channels-last X-ray feature maps feats (B*V, P, E) (P = H'*W' positions) fused with the global
 * average pool over P (reference models/diagnostic_losses.py:126, :131): mean (B, P, E) fp32, pooled (B, E) fp32.
 * bwd: dfeats[(b, v)] = (dmean[b] + dpooled[b] / P) / V for every view (either gradient may be NULL).
 * workspace: hvc_view_mean_gap_workspace floats. */
int64_t hvc_view_mean_gap_workspace(int B, int P, int E);
int hvc_view_mean_gap_fwd(const void* feats, float* mean, float* pooled, float* workspace, int B, int V, int P, int E, int dtype,
                          void* stream);
int hvc_view_mean_gap_bwd(const float* dmean, const float* dpooled, void* dfeats, int B, int V, int P, int E, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HVC_HIP_H */
