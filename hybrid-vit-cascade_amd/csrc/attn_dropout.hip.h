// Attention-probability dropout lots, shared by every attention kernel (attention.hip, attention_fp8.hip): the forward and both
// backward kernels must regenerate exactly the same keep decisions from (seed, b, h, q, k).
#pragma once
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

// ---- attention-probability dropout ------------------------------------------------------------------
// keep(b,h,q,k) = lot16(k & 3 of hash4(rowkey(b,h,q) + (k >> 2) * C)) >= ts.  rowkey is one 32-bit word per
// query row (computed once per lane, or once per LDS tile row in the dKV kernel); one 64-bit multiply of the mixed word
// yields two words of two 16-bit lots each (drop_lots4), i.e. four consecutive keys share the hash.
// Lots are read as SIGNED 16-bit numbers and the threshold is biased accordingly (ts = p * 65536 - 32768):
//  * forward: a saturating packed subtract + packed arithmetic shift turn a word of two lots into a word of
//    two 0x0000 / 0xffff keep masks, ANDed onto the packed bf16 probabilities (1.5 instructions / element);
//  * backward: the high lot is a plain 32-bit signed compare of the whole word against ts << 16, the low
//    lot a 16-bit compare - no field extraction.
// The 1/(1-p) factor is applied once to the accumulators in the epilogues, not per element.
__device__ __forceinline__ uint32_t drop_rowkey(const AttnArgs& a, int bh, int q) {
    return mix32(((uint32_t)(bh * a.Nq + q) * 0x9E3779B1u) ^ a.seed_lo) ^ a.seed_hi;
}
// The word that is multiplied: x = ((rowkey ^ (j & 1 ? H : 0)) + tile * T) ^ A[j >> 3] ^ B[(j >> 1) & 3], tile = key >> 6,
// j = (key >> 2) & 15 the 4-key group inside the 64-key tile, all constants fixed odd words.  Every kernel gets x
// for one instruction per group whatever its register layout: in the query-on-lane kernels j & 1 is the lane half,
// folded into the row key once, and A ^ B is a compile-time constant per group; the dK/dV lot generator keeps
// both parities of its row key per tile with its thread's A term folded in.
constexpr uint32_t kLotMulA = 0x7feb352dU;
constexpr uint32_t kTileAdd = 0x9E3779B9u, kGrpH = 0xB55A4F09u;
__host__ __device__ constexpr uint32_t drop_grp_a(int j8) { return j8 ? 0xC2B2AE35u : 0x85EBCA6Bu; }
__host__ __device__ constexpr uint32_t drop_grp_b(int m) { return m == 0 ? 0x27D4EB2Fu : m == 1 ? 0x165667B1u : m == 2 ? 0xD3A2646Cu : 0xFD7046C5u; }
// Four 16-bit lots from ONE 32 x 32 -> 64-bit multiply (v_mad_u64_u32) instead of two 32-bit multiply-xorshift rounds.
// The low word is xor-shifted (its low bits see only the low bits of the input) and carries lots 0 / 1.  The high word
// only spans [0, M) with M = 0x7feb352d ~ 2^31, so lots 2 / 3 are (high + first word): uniform because the first word is,
// and - the offset being uniform over HALF the range - a drop of lot 2k+1 has probability p / (2 M / 2^32) = 1.0006 p
// given a drop of the lot above it, i.e. the keep events stay pairwise uncorrelated (the mask statistics test checks the
// rate and the correlations inside a hash group on the recovered mask).
__device__ __forceinline__ void drop_lots4(uint32_t mixed, uint32_t& a, uint32_t& b) {
    const uint64_t pr = (uint64_t)mixed * (uint64_t)kLotMulA;
    const uint32_t lo = (uint32_t)pr;
    a = lo ^ (lo >> 15);
    b = (uint32_t)(pr >> 32) + a;
}
__device__ __forceinline__ int drop_ts(const AttnArgs& a) { return (int)a.drop_thresh - 32768; }
__device__ __forceinline__ bool drop_keep_lo(uint32_t w, int ts) { return (int16_t)w >= (int16_t)ts; }
__device__ __forceinline__ bool drop_keep_hi(uint32_t w, int ts) { return (int32_t)w >= ts * 65536; }
// keep flags of the four consecutive keys of one group from its mixed word
__device__ __forceinline__ void drop_keep4(uint32_t mixed, int ts, bool (&keep)[4]) {
    uint32_t a, b;
    drop_lots4(mixed, a, b);
    keep[0] = drop_keep_lo(a, ts); keep[1] = drop_keep_hi(a, ts);
    keep[2] = drop_keep_lo(b, ts); keep[3] = drop_keep_hi(b, ts);
}
// out[e] = keep(e) ? x[e] : alt for the four consecutive keys of one group.  The low lots are tested by a true 16-bit compare
// (v_cmp_ge_i16 reads bits 15:0 of both operands); written as `(int16_t)w >= (int16_t)ts` hipcc canonicalises the test into
// a shift and a 32-bit compare, one more instruction per element pair.  Only the COMPARE is an asm statement (its operands are
// the lot word and the threshold); the select is the compiler's own v_cndmask on the lane mask (inverse ballot), because x is
// a fresh MFMA result in the dQ kernel and hipcc does not pad the MFMA -> VALU read hazard for operands of an asm statement:
// with the select in asm the dQ kernel read stale accumulators on some shapes (round 3: scripts/attn_determinism.py).
__device__ __forceinline__ float drop_select_lo(uint32_t w, int ts, float x, float alt) {
    uint64_t m;
    asm("v_cmp_ge_i16_e64 %0, %1, %2" : "=s"(m) : "v"(w), "v"(ts));
    return __builtin_amdgcn_inverse_ballot_w64(m) ? x : alt;
}
__device__ __forceinline__ void drop_select4(uint32_t mixed, int ts, const float (&x)[4], float alt, float (&out)[4]) {
    uint32_t a, b;
    drop_lots4(mixed, a, b);
    out[0] = drop_select_lo(a, ts, x[0], alt);
    out[1] = drop_keep_hi(a, ts) ? x[1] : alt;
    out[2] = drop_select_lo(b, ts, x[2], alt);
    out[3] = drop_keep_hi(b, ts) ? x[3] : alt;
}
// word of two lots -> word of two 16-bit keep masks (0xffff = keep).  tm1x2 = (ts - 1) in both halves:
// sat(ts - 1 - lot) is negative exactly when lot >= ts, and its sign fills the half.
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t drop_keepmask2(uint32_t lots, uint32_t tm1x2) {
    i16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(i16x2, tm1x2), __builtin_bit_cast(i16x2, lots));
    d = d >> 15;
    return __builtin_bit_cast(uint32_t, d);
}

}  // namespace
}  // namespace hvc
