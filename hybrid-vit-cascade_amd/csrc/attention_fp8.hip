// Attention forward with fp8 (OCP e4m3) MFMA products for gfx950: v_mfma_f32_32x32x16_fp8_fp8 for Q K^T and P V.
// Opt-in variant of the forward in attention.hip (BASELINE configs[4]: "fp8 MFMA attention" of cascade stage 3); same
// mathematics (models/vit_components.py:41-51, :95-113), same dropout lots, fp32 softmax statistics, bf16 in / out:
//   * a pre-pass quantises K to fp8 rows [b*h][Nk][D] and V to fp8 TRANSPOSED rows [b*h][D][Nk64] (Nk64 = Nk rounded up to a
//     64-key tile, zero filled): the P V product then takes V^T row fragments straight from LDS (no 8-bit transposed read);
//   * Q is scaled by scale * log2(e) and converted in registers; scores are therefore in exp2 units, as everywhere;
//   * P is formed as exp2(s - m + 8), i.e. scaled by 2^8: with the running reference kept within 2^0.75 of the row maximum
//     the scaled probabilities lie in (0, 431] of e4m3's [2^-9, 448], so the smallest probability kept is 2^-17 of the
//     largest.  The row sum carries the same factor, so O = (sum P V) / (sum P) needs no correction; LSE subtracts the 8;
//   * e4m3 keeps 3 mantissa bits (rms rounding error 3.7 %): stated tolerance 6e-2 relative Frobenius error on O for white-noise
//     operands (measured 5.3e-2: the P and V roundings in quadrature), 5e-2 on the reference's block fixture (tests/).
// The MFMA count per tile equals the bf16 kernel's (K = 16 per instruction for fp8 as for bf16 at 32x32), so with the
// softmax / dropout vector work bounding the kernel this variant is not faster; it exists so that the
// configuration the reference's baseline names can be run and measured.
#include <cstdlib>
#include "hvc_common.hip.h"
#include "hvc_kernels.h"
#include "attn_dropout.hip.h"

namespace hvc {
namespace {

constexpr int kKT8 = 64;           // keys per LDS tile
constexpr int kQB8 = 128;          // query rows per workgroup (4 waves x 32)
constexpr int kKStride = 72;       // bytes per K tile row in LDS   (64 + 8: the 32 rows a ds_read_b64 group touches land on distinct banks)
constexpr int kVStride = 68;       // bytes per V^T tile row in LDS (64 + 4: same for the ds_read_b32 pairs)
constexpr float kRescale8 = 0.75f; // the running reference stays within 2^0.75 of the row maximum: p * 2^8 <= 431 < 448
constexpr float kPShift = 8.f;

typedef long fp8x8;                // eight e4m3 values: one 32x32x16 MFMA operand

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
__device__ __forceinline__ fp8x8 pack8_fp8(const float (&x)[8]) {
    const uint32_t lo = pack4_fp8(x[0], x[1], x[2], x[3]), hi = pack4_fp8(x[4], x[5], x[6], x[7]);
    return (fp8x8)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ f32x16 mfma_fp8(fp8x8 a, fp8x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
}

// ---- pre-pass: K -> fp8 rows, V -> fp8 transposed rows ------------------------------------------------------------------
// thread -> (key, 8-wide d chunk) with the key fastest, so that the byte stores of a wavefront into one V^T row are contiguous
// MX: the keys of every 64-key V^T tile are stored in the order in which a lane of the 32x32 score accumulators holds them
// (key 32 kt + 8 g + 4 h + j at byte 32 h + 16 kt + 4 g + j), so that the 32 k-slots of lane half h of a 32x32x64 MFMA are one
// contiguous 32-byte run of the V^T row and the matching P^T operand is the lane's own accumulators, packed in register order.
__host__ __device__ constexpr int mx_key_pos(int kk) { return 32 * ((kk >> 2) & 1) + 16 * (kk >> 5) + 4 * ((kk >> 3) & 3) + (kk & 3); }

template <int D, bool MX>
__global__ __launch_bounds__(256) void attn_fp8_prep_kernel(const AttnArgs a, uint8_t* __restrict__ k8, uint8_t* __restrict__ vt8, int nk64) {
    constexpr int CH = D / 8;
    const int64_t total = (int64_t)a.B * a.H * CH * nk64;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int key = (int)(idx % nk64);
        const int ch = (int)((idx / nk64) % CH);
        const int bh = (int)(idx / ((int64_t)nk64 * CH));
        const int b = bh / a.H, hh = bh % a.H;
        float kx[8], vx[8];
        if (key < a.Nk) {
            const bf16* kp = reinterpret_cast<const bf16*>(a.k) + b * a.k_sb + hh * a.k_sh + (int64_t)key * a.k_sn + 8 * ch;
            const bf16* vp = reinterpret_cast<const bf16*>(a.v) + b * a.v_sb + hh * a.v_sh + (int64_t)key * a.v_sn + 8 * ch;
            const bf16x8 kk = *reinterpret_cast<const bf16x8*>(kp), vv = *reinterpret_cast<const bf16x8*>(vp);
#pragma unroll
            for (int j = 0; j < 8; ++j) { kx[j] = bf2f(kk[j]); vx[j] = bf2f(vv[j]); }
            *reinterpret_cast<fp8x8*>(k8 + ((int64_t)bh * a.Nk + key) * D + 8 * ch) = pack8_fp8(kx);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) vx[j] = 0.f;
        }
        const fp8x8 vq = pack8_fp8(vx);
#pragma unroll
        for (int j = 0; j < 8; ++j) vt8[((int64_t)bh * D + 8 * ch + j) * nk64 + (MX ? (key & ~63) + mx_key_pos(key & 63) : key)] = (uint8_t)((uint64_t)vq >> (8 * j));
    }
}

// ---- forward -----------------------------------------------------------------------------------------------------------
template <int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(const AttnArgs a_in, const uint8_t* __restrict__ k8, const uint8_t* __restrict__ vt8, int nk64) {
    AttnArgs a = a_in;
    a.seed_lo = seed_with_counter(a_in.seed_lo, a_in.seed_ctr);
    constexpr int DS = D / 16, DT = D / 32;
    constexpr int KT_BYTES = kKT8 * kKStride, VT_BYTES = D * kVStride;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [buf][K tile | V^T tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int nqb = (a.Nq + kQB8 - 1) / kQB8;
    int bh, qb;
    if ((a.B * a.H & 7) == 0) {          // keep one (b, h)'s query blocks on one XCD (speed only), as attention.hip
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = xcd + 8 * (slot / nqb);
        qb = slot % nqb;
    } else {
        bh = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    const int b = bh / a.H, hh = bh % a.H;
    const bf16* qp = reinterpret_cast<const bf16*>(a.q) + b * a.q_sb + hh * a.q_sh;
    bf16* op = reinterpret_cast<bf16*>(a.o) + b * a.o_sb + hh * a.o_sh;
    const uint8_t* kb = k8 + (int64_t)bh * a.Nk * D;
    const uint8_t* vb = vt8 + (int64_t)bh * D * nk64;

    const int qrow = qb * kQB8 + wave * 32 + r;
    const bool qvalid = qrow < a.Nq;
    const int qrow_c = qvalid ? qrow : a.Nq - 1;
    const float sl2 = a.scale * kLog2e;
    fp8x8 qf[DS];                                    // B operand: (c Q)^T, lane = query column, k = d
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        const bf16x8 qq = *reinterpret_cast<const bf16x8*>(qp + (int64_t)qrow_c * a.q_sn + 16 * s + 8 * h);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = bf2f(qq[j]) * sl2;
        qf[s] = pack8_fp8(x);
    }

    // tile loaders: 16-byte chunks; K tile = 64 rows x D bytes, V^T tile = D rows x 64 bytes
    constexpr int KCH = kKT8 * D / 16, VCH = D * 64 / 16;             // chunks per tile (256 / 128 for D = 64 / 32)
    const bool kact = tid < KCH, vact = tid < VCH;
    const int krow = tid / (D / 16), kch = tid % (D / 16);            // K: row = key, chunk of 16 d
    const int vrow = tid / 4, vch = tid % 4;                          // V^T: row = d, chunk of 16 keys
    u32x4 kreg = {0, 0, 0, 0}, vreg = {0, 0, 0, 0};
    auto issue = [&](int t) {
        const int key0 = t * kKT8;
        if (kact) {
            const int key = key0 + krow;
            kreg = key < a.Nk ? *reinterpret_cast<const u32x4*>(kb + (int64_t)key * D + 16 * kch) : u32x4{0, 0, 0, 0};
        }
        if (vact) vreg = *reinterpret_cast<const u32x4*>(vb + (int64_t)vrow * nk64 + key0 + 16 * vch);      // zero padded to nk64
    };
    auto commit = [&](int buf) {
        char* kt = smem + buf * (KT_BYTES + VT_BYTES);
        char* vt = kt + KT_BYTES;
        if (kact) {
            uint64_t* dst = reinterpret_cast<uint64_t*>(kt + krow * kKStride + 16 * kch);      // 8-byte aligned (72 r + 16 c)
            dst[0] = ((uint64_t)kreg[1] << 32) | kreg[0];
            dst[1] = ((uint64_t)kreg[3] << 32) | kreg[2];
        }
        if (vact) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(vt + vrow * kVStride + 16 * vch);      // 4-byte aligned (68 r + 16 c)
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i] = vreg[i];
        }
    };
    const int nt = (a.Nk + kKT8 - 1) / kKT8;
    issue(0);
    commit(0);
    __syncthreads();

    f32x16 negm, o[DT];
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = kPShift;                   // accumulator seed: -reference + 8
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    const uint32_t rowkey = DROP ? drop_rowkey(a, bh, qrow_c) ^ (h ? kGrpH : 0u) : 0u;
    const int ts = drop_ts(a);

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const char* kt_ = smem + buf * (KT_BYTES + VT_BYTES);
        const char* vt_ = kt_ + KT_BYTES;
        if (t + 1 < nt) issue(t + 1);
        // S^T[key][q] - ref[q] + 8 = K (c Q)^T + seed
        f32x16 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[kt] = negm;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const fp8x8 kf = *reinterpret_cast<const fp8x8*>(kt_ + (32 * kt + r) * kKStride + 16 * s + 8 * h);
                st[kt] = mfma_fp8(kf, qf[s], st[kt]);
            }
        }
        const int kbase = t * kKT8;
        if (kbase + kKT8 > a.Nk) {                                    // ragged last tile
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kbase + 32 * kt + acc_row(i, h) >= a.Nk) st[kt][i] = -INFINITY;
        }
        float mloc = -INFINITY;                                       // row maximum relative to (reference - 8)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, st[kt][i]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (t == 0 || __any(mloc > kPShift + kRescale8)) {            // wave-uniform; O, l, the seed and this tile's scores move together
            const float shift = t == 0 ? mloc - kPShift : fmaxf(mloc - kPShift, 0.f);
            const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-shift);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            const float nm = negm[0] - shift;
#pragma unroll
            for (int i = 0; i < 16; ++i) negm[i] = nm;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) st[kt][i] -= shift;
        }
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(st[kt][i]);
                rs += p;
                st[kt][i] = p;
            }
        l += rs;
        if constexpr (DROP) {                                         // same lots as attention.hip (drop_lots4 of the group word)
            const uint32_t rk_tile = rowkey + (uint32_t)t * kTileAdd;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    bool keep[4];
                    drop_keep4(rk_tile ^ (drop_grp_a(kt) ^ drop_grp_b(i >> 2)), ts, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!keep[j]) st[kt][i + j] = 0.f;
                }
        }
        // O^T[d][q] += V^T P^T: A = V^T row fragment in the k order of the accumulator block, B = P^T from the accumulators
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = st[kt][8 * s2 + j];
                const fp8x8 pf = pack8_fp8(x);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* row = vt_ + (32 * dt + r) * kVStride + 32 * kt + 16 * s2 + 4 * h;
                    const uint32_t lo = *reinterpret_cast<const uint32_t*>(row), hi = *reinterpret_cast<const uint32_t*>(row + 8);
                    o[dt] = mfma_fp8((fp8x8)(((uint64_t)hi << 32) | lo), pf, o[dt]);
                }
            }
        if (t + 1 < nt) commit(buf ^ 1);
        __syncthreads();
    }

    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = (DROP ? a.keep_scale : 1.f) / ltot;
    if (qvalid) {
        bf16* orow = op + (int64_t)qrow * a.o_sn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = f2bf(o[dt][4 * g + j] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * dt + 8 * g + 4 * h) = w;
            }
        // negm = -reference + 8, l carries 2^8: lse2 = reference + log2(l) - 8
        if (h == 0) a.lse[(int64_t)bh * a.Nq + qrow] = (__builtin_amdgcn_logf(ltot) - negm[0]) * kLn2;
    }
}

// ---- forward, experiment of round 3 ---------------------------------------------------------------------
// Same mathematics and lots as attn_fwd_fp8_kernel with two changes:
//   * P V (and Q K^T at d = 64) go through v_mfma_f32_32x32x64_f8f6f4 (the block-scaled instruction's unit-scale form: hipcc
//     lowers __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4 with zero scale operands to it): one instruction per 64 keys (or 64
//     head channels) instead of four 32x32x16, at twice the rate;
//   * the steady state takes no row maximum: probabilities are formed against the current reference as exp2(s - ref + 3) and the
//     lane's sum of its 32 probabilities (needed anyway) votes: a sum <= 448 proves every probability fits e4m3; otherwise (and
//     on the first tile) the scores are rebuilt from LDS and the reference moves to the row maximum.  The shift is 3, not 8: a
//     flat row (every p = 2^3) sums to 256 per lane and must not vote; the smallest probability kept is 2^-12 of the reference.
constexpr int kStrideMX = 80;      // bytes per K / V^T tile row in LDS: 16-byte aligned rows, the 8 rows of a ds_read_b128 group on distinct banks
constexpr float kPShiftMX = 3.f;
constexpr float kSumCeilMX = 448.f;
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8mx_kernel(const AttnArgs a_in, const uint8_t* __restrict__ k8, const uint8_t* __restrict__ vt8, int nk64) {
    AttnArgs a = a_in;
    a.seed_lo = seed_with_counter(a_in.seed_lo, a_in.seed_ctr);
    constexpr int DS = D / 16, DT = D / 32;
    constexpr int KT_BYTES = kKT8 * kStrideMX, VT_BYTES = D * kStrideMX;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [buf][K tile | V^T tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int nqb = (a.Nq + kQB8 - 1) / kQB8;
    int bh, qb;
    if ((a.B * a.H & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = xcd + 8 * (slot / nqb);
        qb = slot % nqb;
    } else {
        bh = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    const int b = bh / a.H, hh = bh % a.H;
    const bf16* qp = reinterpret_cast<const bf16*>(a.q) + b * a.q_sb + hh * a.q_sh;
    bf16* op = reinterpret_cast<bf16*>(a.o) + b * a.o_sb + hh * a.o_sh;
    const uint8_t* kb = k8 + (int64_t)bh * a.Nk * D;
    const uint8_t* vb = vt8 + (int64_t)bh * D * nk64;

    const int qrow = qb * kQB8 + wave * 32 + r;
    const bool qvalid = qrow < a.Nq;
    const int qrow_c = qvalid ? qrow : a.Nq - 1;
    const float sl2 = a.scale * kLog2e;
    // B operand (c Q)^T.  d = 32: two 32x32x16 fragments (d = 16 s + 8 h ..); d = 64: one 32x32x64 fragment, lane half h holds d = 32 h .. 32 h + 31
    fp8x8 qf[DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        const int d0 = D == 64 ? 32 * h + 8 * s : 16 * s + 8 * h;
        const bf16x8 qq = *reinterpret_cast<const bf16x8*>(qp + (int64_t)qrow_c * a.q_sn + d0);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = bf2f(qq[j]) * sl2;
        qf[s] = pack8_fp8(x);
    }

    constexpr int KCH = kKT8 * D / 16, VCH = D * 64 / 16;
    const bool kact = tid < KCH, vact = tid < VCH;
    const int krow = tid / (D / 16), kch = tid % (D / 16);
    const int vrow = tid / 4, vch = tid % 4;
    u32x4 kreg = {0, 0, 0, 0}, vreg = {0, 0, 0, 0};
    auto issue = [&](int t) {
        const int key0 = t * kKT8;
        if (kact) {
            const int key = key0 + krow;
            kreg = key < a.Nk ? *reinterpret_cast<const u32x4*>(kb + (int64_t)key * D + 16 * kch) : u32x4{0, 0, 0, 0};
        }
        if (vact) vreg = *reinterpret_cast<const u32x4*>(vb + (int64_t)vrow * nk64 + key0 + 16 * vch);
    };
    auto commit = [&](int buf) {
        char* kt = smem + buf * (KT_BYTES + VT_BYTES);
        char* vt = kt + KT_BYTES;
        if (kact) *reinterpret_cast<u32x4*>(kt + krow * kStrideMX + 16 * kch) = kreg;
        if (vact) *reinterpret_cast<u32x4*>(vt + vrow * kStrideMX + 16 * vch) = vreg;
    };
    const int nt = (a.Nk + kKT8 - 1) / kKT8;
    issue(0);
    commit(0);
    __syncthreads();

    f32x16 negm, o[DT];
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = kPShiftMX;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    const uint32_t rowkey = DROP ? drop_rowkey(a, bh, qrow_c) ^ (h ? kGrpH : 0u) : 0u;
    const int ts = drop_ts(a);

    // Two-level loop: the inner loop is the steady state and changes neither the reference nor the scale of O and l, so that its
    // registers carry no merge copies; a vote leaves it with the tile's scores rebuilt and the outer loop moves the reference.
    f32x16 st[2];
    float rs;
    auto scores = [&](int t) {                                        // S^T[key][q] - ref[q] + shift = K (c Q)^T + seed
        const char* kt_ = smem + (t & 1) * (KT_BYTES + VT_BYTES);
        const int kbase = t * kKT8;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[kt] = negm;
            if constexpr (D == 64) {
                const char* row = kt_ + (32 * kt + r) * kStrideMX + 32 * h;
                const u32x4 lo = *reinterpret_cast<const u32x4*>(row), hi = *reinterpret_cast<const u32x4*>(row + 16);
                const i32x8 kf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                i32x8 qv;
#pragma unroll
                for (int s = 0; s < 4; ++s) { qv[2 * s] = (int)(uint32_t)(uint64_t)qf[s]; qv[2 * s + 1] = (int)(uint32_t)((uint64_t)qf[s] >> 32); }
                st[kt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qv, st[kt], 0, 0, 0, 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < DS; ++s) {
                    const fp8x8 kf = *reinterpret_cast<const fp8x8*>(kt_ + (32 * kt + r) * kStrideMX + 16 * s + 8 * h);
                    st[kt] = mfma_fp8(kf, qf[s], st[kt]);
                }
            }
        }
        if (kbase + kKT8 > a.Nk) {                                    // ragged last tile
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kbase + 32 * kt + acc_row(i, h) >= a.Nk) st[kt][i] = -INFINITY;
        }
    };
    auto exps = [&]() {
        float acc0 = 0.f, acc1 = 0.f;                                 // two plain add chains (see attention.hip: no dependent v_pk_add_f32 chain)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const float p0 = __builtin_amdgcn_exp2f(st[kt][i]), p1 = __builtin_amdgcn_exp2f(st[kt][i + 1]);
                acc0 += p0;
                asm("" : "+v"(acc0));
                acc1 += p1;
                asm("" : "+v"(acc1));
                st[kt][i] = p0;
                st[kt][i + 1] = p1;
            }
        rs = acc0 + acc1;
    };
    auto move_reference = [&](int t) {                                // scores of tile t in st; moves the reference to the row maximum
        float mloc = -INFINITY;                                       // row maximum relative to (reference - shift)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, st[kt][i]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float shift = t == 0 ? mloc - kPShiftMX : fmaxf(mloc - kPShiftMX, 0.f);
        const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-shift);
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        const float nm = negm[0] - shift;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = nm;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kt][i] -= shift;
    };
    auto finish = [&](int t) {                                        // probabilities of tile t in st, their lane sum in rs
        const char* vt_ = smem + (t & 1) * (KT_BYTES + VT_BYTES) + KT_BYTES;
        l += rs;
        if constexpr (DROP) {
            const uint32_t rk_tile = rowkey + (uint32_t)t * kTileAdd;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    bool keep[4];
                    drop_keep4(rk_tile ^ (drop_grp_a(kt) ^ drop_grp_b(i >> 2)), ts, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!keep[j]) st[kt][i + j] = 0.f;
                }
        }
        // O^T[d][q] += V^T P^T, 64 keys per instruction: B = the lane's 32 probabilities in register order (k-slot 16 kt + i),
        // A = the 32 bytes of V^T row d at 32 h (the tile is stored in that key order: mx_key_pos)
        i32x8 pv;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) pv[4 * kt + g] = (int)pack4_fp8(st[kt][4 * g], st[kt][4 * g + 1], st[kt][4 * g + 2], st[kt][4 * g + 3]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const char* row = vt_ + (32 * dt + r) * kStrideMX + 32 * h;
            const u32x4 lo = *reinterpret_cast<const u32x4*>(row), hi = *reinterpret_cast<const u32x4*>(row + 16);
            const i32x8 vf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            o[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pv, o[dt], 0, 0, 0, 0, 0, 0);
        }
        if (t + 1 < nt) commit((t & 1) ^ 1);
        __syncthreads();
    };
    int t = 0;
    bool have_scores = false;                                         // tile t's loads issued and its scores in st (left by a vote)
    while (t < nt) {
        if (!have_scores) {
            if (t + 1 < nt) issue(t + 1);
            scores(t);
        }
        move_reference(t);
        exps();
        finish(t);
        have_scores = false;
        for (++t; t < nt; ++t) {
            if (t + 1 < nt) issue(t + 1);
            scores(t);
            exps();
            if (__any(!(rs <= kSumCeilMX))) {                         // some probability may not fit e4m3 (inf and NaN vote too)
                scores(t);
                have_scores = true;
                break;
            }
            finish(t);
        }
    }

    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = (DROP ? a.keep_scale : 1.f) / ltot;
    if (qvalid) {
        bf16* orow = op + (int64_t)qrow * a.o_sn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = f2bf(o[dt][4 * g + j] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * dt + 8 * g + 4 * h) = w;
            }
        if (h == 0) a.lse[(int64_t)bh * a.Nq + qrow] = (__builtin_amdgcn_logf(ltot) - negm[0]) * kLn2;
    }
}

}  // namespace

int64_t attention_fp8_workspace_bytes(int B, int H, int Nk, int D) {
    const int64_t nk64 = ((int64_t)Nk + 63) / 64 * 64;
    return (int64_t)B * H * ((int64_t)Nk * D + (int64_t)D * nk64);
}

// HVC_FP8_MX=1 selects the round-3 experiment kernel (32x32x64 products, sum-voted reference); read at every launch so that
// one process (the parity tests) can run both
static bool fp8_mx() {
    return option(kOptFp8Mx) == 1;
}

template <int D>
static hipError_t launch_fp8(const AttnArgs& a, uint8_t* ws, hipStream_t st) {
    const int nk64 = (a.Nk + 63) / 64 * 64;
    uint8_t* k8 = ws;
    uint8_t* vt8 = ws + (int64_t)a.B * a.H * a.Nk * D;
    const int64_t work = (int64_t)a.B * a.H * (D / 8) * nk64;
    int blocks = (int)((work + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    const bool mx = fp8_mx();
    if (mx) hipLaunchKernelGGL((attn_fp8_prep_kernel<D, true>), dim3(blocks), dim3(256), 0, st, a, k8, vt8, nk64);
    else hipLaunchKernelGGL((attn_fp8_prep_kernel<D, false>), dim3(blocks), dim3(256), 0, st, a, k8, vt8, nk64);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nqb = (a.Nq + kQB8 - 1) / kQB8;
    const dim3 grid(nqb * a.B * a.H), block(256);
    if (mx) {
        const size_t lds = (size_t)2 * (kKT8 + D) * kStrideMX;
        if (a.drop_thresh) hipLaunchKernelGGL((attn_fwd_fp8mx_kernel<D, true>), grid, block, lds, st, a, k8, vt8, nk64);
        else hipLaunchKernelGGL((attn_fwd_fp8mx_kernel<D, false>), grid, block, lds, st, a, k8, vt8, nk64);
        return hipGetLastError();
    }
    const size_t lds = (size_t)2 * (kKT8 * kKStride + D * kVStride);
    if (a.drop_thresh) hipLaunchKernelGGL((attn_fwd_fp8_kernel<D, true>), grid, block, lds, st, a, k8, vt8, nk64);
    else hipLaunchKernelGGL((attn_fwd_fp8_kernel<D, false>), grid, block, lds, st, a, k8, vt8, nk64);
    return hipGetLastError();
}

hipError_t attention_fp8_launch(const AttnArgs& a, void* workspace, hipStream_t st) {
    if (a.D == 64) return launch_fp8<64>(a, (uint8_t*)workspace, st);
    if (a.D == 32) return launch_fp8<32>(a, (uint8_t*)workspace, st);
    return hipErrorInvalidValue;
}

}  // namespace hvc
