// Attention-probability dropout lots, shared by every attention kernel (attention.hip, attention_fp8.hip): the forward and both
// backward kernels must regenerate exactly the same keep decisions from (seed, b, h, q, k).
#pragma once
#include "hvc_common.hip.h"
#include "hvc_kernels.h"

namespace hvc {
namespace {

// ---- attention-probability dropout ------------------------------------------------------------------
// keep(b,h,q,k) = lot16(k & 3 of hash4(rowkey(b,h,q) + (k >> 2) * C)) >= ts.  rowkey is one 32-bit word per
// query row (computed once per lane, or once per LDS tile row in the dKV kernel); two high-half multiplies of the mixed word
// yield two words of two 16-bit lots each (drop_lots4), i.e. four consecutive keys share the hash.
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
constexpr uint32_t kLotMulA = 0xD6E8FEB9u, kLotMulB = 0xA54FF53Bu;
constexpr uint32_t kTileAdd = 0x9E3779B9u, kGrpH = 0xB55A4F09u;
__host__ __device__ constexpr uint32_t drop_grp_a(int j8) { return j8 ? 0xC2B2AE35u : 0x85EBCA6Bu; }
__host__ __device__ constexpr uint32_t drop_grp_b(int m) { return m == 0 ? 0x27D4EB2Fu : m == 1 ? 0x165667B1u : m == 2 ? 0xD3A2646Cu : 0xFD7046C5u; }
// Four 16-bit lots from TWO high-half multiplies (v_mul_hi_u32) of the mixed word by two unrelated odd constants: bits 32..63
// of a product depend on every bit of the input, so all four lots are well mixed with no xor-shift round (3 instructions per
// group with the mixing xor; the previous form - one 64-bit multiply, xor-shift of the low word, add - took 5).
// The low lot of each word (product bits 32..47) is uniform on 16 bits.  The high lot (bits 48..63) only spans [0, M >> 16],
// uniformly, so it is tested against its own threshold T_M = T * M / 2^32 (same drop probability T / 65536); the drop window
// [32768, 32768 + T_M) of the raw lot lies inside that range for M > 0x9999ffff.
__device__ __forceinline__ void drop_lots4(uint32_t mixed, uint32_t& a, uint32_t& b) {
    a = __umulhi(mixed, kLotMulA);
    b = __umulhi(mixed, kLotMulB);
}
// Signed-lot thresholds: a lot is dropped when (int16)lot < ts.  lo: the two low lots; hia / hib: the high lots of words a / b.
struct DropTs { int lo, hia, hib; };
__device__ __forceinline__ DropTs drop_ts(const AttnArgs& a) {
    DropTs t;
    const uint64_t T = a.drop_thresh;
    t.lo = (int)T - 32768;
    t.hia = (int)((T * kLotMulA + 0x80000000ull) >> 32) - 32768;
    t.hib = (int)((T * kLotMulB + 0x80000000ull) >> 32) - 32768;
    return t;
}
__device__ __forceinline__ bool drop_keep_lo(uint32_t w, int ts) { return (int16_t)w >= (int16_t)ts; }
__device__ __forceinline__ bool drop_keep_hi(uint32_t w, int ts) { return (int32_t)w >= ts * 65536; }
// keep flags of the four consecutive keys of one group from its mixed word
__device__ __forceinline__ void drop_keep4(uint32_t mixed, const DropTs& ts, bool (&keep)[4]) {
    uint32_t a, b;
    drop_lots4(mixed, a, b);
    keep[0] = drop_keep_lo(a, ts.lo); keep[1] = drop_keep_hi(a, ts.hia);
    keep[2] = drop_keep_lo(b, ts.lo); keep[3] = drop_keep_hi(b, ts.hib);
}
// word of two lots -> word of two 16-bit keep masks (0xffff = keep).  tm1x2 = (ts - 1) of the low lot in the low half and of
// the word's high lot in the high half (drop_tm1x2): sat(ts - 1 - lot) is negative exactly when lot >= ts, and its sign
// fills the half.
__device__ __forceinline__ uint32_t drop_tm1x2(int ts_lo, int ts_hi) {
    return ((uint32_t)(ts_lo - 1) & 0xffffu) | (((uint32_t)(ts_hi - 1) & 0xffffu) << 16);
}
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t drop_keepmask2(uint32_t lots, uint32_t tm1x2) {
    i16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(i16x2, tm1x2), __builtin_bit_cast(i16x2, lots));
    d = d >> 15;
    return __builtin_bit_cast(uint32_t, d);
}

}  // namespace
}  // namespace hvc
