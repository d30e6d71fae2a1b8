// Device-side building blocks shared by every gfx950 kernel in this library.
//
// Everything here is written for CDNA4 only: 64-lane wavefronts,
// v_mfma_f32_32x32x16_bf16, ds_read_b64_tr_b16, 160 KiB LDS per CU.
//
// Fragment conventions (MI355X, mfma_f32_32x32x16_bf16; lane l, r = l & 31, h = l >> 5):
//   A operand : A[row r][k = 8h + j]      j = 0..7  (bf16x8)
//   B operand : B[k = 8h + j][col r]      j = 0..7  (bf16x8)
//   C/D       : C[row (i&3) + 8(i>>2) + 4h][col r]   i = 0..15 (f32x16)
// so both operands are "8 consecutive k for my r", i.e. C[i][j] = sum_k A'[i][k] * B'[j][k]
// with A', B' stored k-contiguous is the native form ("row fragment"); an operand whose
// contraction index is its *row* index in memory is fetched with the hardware transposed
// LDS read ("tr fragment").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvc {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Number of bf16 images an operand of type T is carried in.  fp32 operands are split
// into hi + lo bf16 parts and every product is evaluated as hi*hi + hi*lo + lo*hi
// (relative error ~2^-16 per product, fp32 accumulate), so the "fp32 mode" of every
// MFMA kernel runs through exactly the same tiles, fragments and schedules as bf16 mode.
template <typename T> struct NSplit { static constexpr int value = 1; };
template <> struct NSplit<float> { static constexpr int value = 2; };

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Row index inside a 32x32 accumulator tile of register i for lane-half h.
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------
// 8-element global loads that yield the bf16 image(s) of the operand.
// ---------------------------------------------------------------------------------
template <typename T> struct Chunk8;   // raw 8 elements as they sit in HBM
template <> struct Chunk8<bf16> { bf16x8 v; };
template <> struct Chunk8<float> { f32x4 a, b; };

template <typename T>
__device__ __forceinline__ Chunk8<T> zero_chunk() {
    Chunk8<T> c;
    if constexpr (sizeof(T) == 2) { c.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    else { c.a = f32x4{0, 0, 0, 0}; c.b = f32x4{0, 0, 0, 0}; }
    return c;
}

// p must be 16-byte aligned when vec is true.
template <typename T>
__device__ __forceinline__ Chunk8<T> load_chunk(const T* p, int nvalid, bool vec) {
    Chunk8<T> c = zero_chunk<T>();
    if (nvalid <= 0) return c;
    if (vec && nvalid >= 8) {
        if constexpr (sizeof(T) == 2) { c.v = *reinterpret_cast<const bf16x8*>(p); }
        else { c.a = *reinterpret_cast<const f32x4*>(p); c.b = *reinterpret_cast<const f32x4*>(p + 4); }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nvalid) {
                if constexpr (sizeof(T) == 2) c.v[j] = p[j];
                else { if (j < 4) c.a[j] = p[j]; else c.b[j - 4] = p[j]; }
            }
        }
    }
    return c;
}

template <typename T>
__device__ __forceinline__ float chunk_get(const Chunk8<T>& c, int j) {
    if constexpr (sizeof(T) == 2) return bf2f(c.v[j]);
    else return j < 4 ? c.a[j] : c.b[j - 4];
}

// hi / lo bf16 images of a chunk (lo only meaningful for fp32).
template <typename T>
__device__ __forceinline__ void chunk_split(const Chunk8<T>& c, bf16x8 (&out)[NSplit<T>::value]) {
    if constexpr (sizeof(T) == 2) { out[0] = c.v; }
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = j < 4 ? c.a[j] : c.b[j - 4];
            bf16 hi = f2bf(x);
            out[0][j] = hi;
            out[1][j] = f2bf(x - bf2f(hi));
        }
    }
}

// hi / lo images of 8 accumulator values (used when a product's result feeds the next MFMA).
template <int NS>
__device__ __forceinline__ void acc_split(const float (&x)[8], bf16x8 (&out)[NS]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bf16 hi = f2bf(x[j]);
        out[0][j] = hi;
        if constexpr (NS == 2) out[1][j] = f2bf(x[j] - bf2f(hi));
    }
}

// ---------------------------------------------------------------------------------
// LDS tile image: R rows of CW bf16 (CW = 32 / 64 / 128 => 64 / 128 / 256-byte rows) with the
// 16-byte chunks of each row XOR-swizzled.
// ---------------------------------------------------------------------------------
template <int CW>
__device__ __forceinline__ int tile_off(int row, int chunk) {   // offset in bf16 elements
    // One swizzle per row width serves both read kinds (MI355X LDS: banks = (addr/4) % 64):
    //  * ds_read_b128 row fragments: the 16 rows of a lane group must land on 16 distinct 16-byte slots
    //  * ds_read_b64_tr_b16: the 4 consecutive rows x 4 consecutive chunks a 32-lane half touches
    //    must land on 16 distinct slots as well.
    int sw;
    if constexpr (CW == 128 || CW == 256) {   // 256 / 512-byte rows: every row starts on bank 0
        sw = chunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
    } else if constexpr (CW == 64) {    // 128-byte rows: 2 rows per bank row
        sw = chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
    } else {                            // 64-byte rows: 4 rows per bank row
        static_assert(CW == 32, "tile width must be 32, 64, 128 or 256 bf16");
        sw = chunk ^ ((row >> 2) & 3);
    }
    return row * CW + sw * 8;
}

template <int CW>
__device__ __forceinline__ void tile_store(bf16* tile, int row, int chunk, bf16x8 v) {
    *reinterpret_cast<bf16x8*>(tile + tile_off<CW>(row, chunk)) = v;
}

// Row fragment: elements k0 + 8h .. k0 + 8h + 7 of row r0 + r (k0 multiple of 16).
template <int CW>
__device__ __forceinline__ bf16x8 row_frag(const bf16* tile, int r0, int k0, int lane) {
    int r = r0 + (lane & 31), h = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(tile + tile_off<CW>(r, (k0 >> 3) + h));
}

__device__ __forceinline__ bf16x4 lds_tr4(const bf16* p) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(bf16x4, t);
}

// Transposed fragment of a row-major tile whose ROW index is the contraction index.
// Lane (r, h) receives column c0 + r at rows  kbase + 4h*HS + {0..3}  and  the same + 8/4.
//   PERM = true  : rows k0 + 4h + {0..3} and k0 + 8 + 4h + {0..3}  -- matches the k order of
//                  an accumulator tile re-used as the other operand (registers 8s..8s+7).
//   PERM = false : rows k0 + 8h + {0..7}                           -- natural k order.
// ds_read_b64_tr_b16 semantics: within each group of 16 lanes, lane 4q+p supplies the address
// of row q, columns 4p..4p+3 of a 4x16 block; lane i receives column i, rows 0..3.
// Requires EXEC = all ones.
template <int CW, bool PERM>
__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int k0, int c0, int lane) {
    int g = lane >> 4, i = lane & 15, h = g >> 1;
    int col = c0 + 16 * (g & 1) + 4 * (i & 3);
    int rowa = k0 + (PERM ? 4 * h : 8 * h) + (i >> 2);
    int rowb = rowa + (PERM ? 8 : 4);
    const bf16* pa = tile + tile_off<CW>(rowa, col >> 3) + (col & 7);
    const bf16* pb = tile + tile_off<CW>(rowb, col >> 3) + (col & 7);
    bf16x4 lo = lds_tr4(pa);
    bf16x4 hi = lds_tr4(pb);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Lane-resident LDS addresses: the XOR swizzle of tile_off<32|64> depends on row bits 1..3 only, so for rows
// r0 + lane part with r0 a multiple of 16 the swizzled offset splits into a per-lane part (computed once, kept in
// a register) and a compile-time part r0 * CW (folded into the instruction's immediate offset).  These return the
// per-lane parts; the caller adds tile base and r0 * CW as constants.
template <int CW>
__device__ __forceinline__ int row_frag_lane_off(int k0, int lane) {       // row_frag<CW>(tile, r0, k0, lane) minus r0 * CW
    static_assert(CW == 32 || CW == 64, "row bits above 3 must not enter the swizzle");
    return tile_off<CW>(lane & 31, (k0 >> 3) + (lane >> 5));
}
template <int CW, bool PERM>
__device__ __forceinline__ void tr_frag_lane_off(int c0, int lane, int& offa, int& offb) {   // tr_frag<CW,PERM>(tile, k0, c0, lane) minus k0 * CW
    static_assert(CW == 32 || CW == 64, "row bits above 3 must not enter the swizzle");
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
    const int rowa = (PERM ? 4 * h : 8 * h) + (i >> 2);
    const int rowb = rowa + (PERM ? 8 : 4);
    offa = tile_off<CW>(rowa, col >> 3) + (col & 7);
    offb = tile_off<CW>(rowb, col >> 3) + (col & 7);
}
__device__ __forceinline__ bf16x8 tr_frag_at(const bf16* pa, const bf16* pb) {
    bf16x4 lo = lds_tr4(pa);
    bf16x4 hi = lds_tr4(pb);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// ---------------------------------------------------------------------------------
// Counter-based dropout RNG shared by forward and backward kernels: keep(...) is a pure function of
// (seed, element coordinates); 16-bit thresholds.
// ---------------------------------------------------------------------------------
// Dropout seeds may carry a device-resident step counter (hvc_set_seed_counter): a captured hipGraph replays the same kernel
// arguments every step, so the per-step variation of every dropout mask comes from one word in HBM that the graph's first node
// advances.  Forward and backward of one step read the same value, so they regenerate the same masks.
__device__ __forceinline__ uint32_t seed_with_counter(uint32_t seed_lo, const uint32_t* ctr) {
    return ctr ? seed_lo + *ctr * 0x9E3779B9u : seed_lo;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// Dropout of a row-major (rows x cols) activation (GEMM-epilogue dropouts and their backward): one well-mixed
// 32-bit key per row, then one multiply-xorshift round per 4 consecutive columns yielding four 16-bit lots
// (element kept when lot >= thresh).  ~3 instructions / element when a thread owns 8 consecutive columns,
// against ~20 for a full hash per element.
constexpr uint32_t kDropKeyMul = 0x85EBCA6Bu, kDropLotMulA = 0x7feb352dU, kDropLotMulB = 0x846ca68bU;
__device__ __forceinline__ uint32_t drop2d_rowkey(uint32_t seed_lo, uint32_t seed_hi, uint64_t row) {
    return mix32(((uint32_t)row * 0x9E3779B1u) ^ ((uint32_t)(row >> 32) * 0xC2B2AE35u) ^ seed_lo) ^ seed_hi;
}
// lots of columns 4*col4 .. 4*col4+3: wa = {c0 (low half), c1}, wb = {c2, c3}
__device__ __forceinline__ void drop2d_lots4(uint32_t rowkey, uint32_t col4, uint32_t& wa, uint32_t& wb) {
    uint32_t x = rowkey + col4 * kDropKeyMul;
    x ^= x >> 16;
    wa = x * kDropLotMulA; wa ^= wa >> 15;
    wb = x * kDropLotMulB; wb ^= wb >> 15;
}
__device__ __forceinline__ bool drop2d_keep(uint32_t rowkey, uint32_t col, uint32_t thresh) {   // single element (ragged paths)
    uint32_t wa, wb;
    drop2d_lots4(rowkey, col >> 2, wa, wb);
    const uint32_t w = (col & 2) ? wb : wa;
    return ((w >> (16 * (col & 1))) & 0xffffu) >= thresh;
}
// keep flags of 8 consecutive columns starting at col (col % 8 == 0... any multiple of 4)
__device__ __forceinline__ void drop2d_keep8(uint32_t rowkey, uint32_t col, uint32_t thresh, bool (&keep)[8]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        uint32_t wa, wb;
        drop2d_lots4(rowkey, (col >> 2) + g, wa, wb);
        keep[4 * g + 0] = (wa & 0xffffu) >= thresh; keep[4 * g + 1] = (wa >> 16) >= thresh;
        keep[4 * g + 2] = (wb & 0xffffu) >= thresh; keep[4 * g + 3] = (wb >> 16) >= thresh;
    }
}

// Exact-GELU pieces (reference: nn.GELU() = x * Phi(x), erf form).  Phi through Abramowitz-Stegun 7.1.26
// (|erf error| <= 1.5e-7, i.e. fp32 rounding level): one v_rcp, one v_exp and a degree-5 Horner, no branches;
// exp(-x^2/2) is shared with the density phi(x) needed by the derivative.  ~15 issue slots against ~35 for erff().
__device__ __forceinline__ void gelu_parts(float x, float& Phi, float& phi) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);        // exp(-x^2 / 2)
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float half_erfc = 0.5f * t * poly * e;                                    // 0.5 * erfc(|x| / sqrt 2)
    Phi = x >= 0.f ? 1.f - half_erfc : half_erfc;
    phi = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float Phi, phi;
    gelu_parts(x, Phi, phi);
    return x * Phi;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float Phi, phi;
    gelu_parts(x, Phi, phi);
    return fmaf(x, phi, Phi);
}

// wave-level sum over all 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Second stage of the deterministic two-stage column reductions: sum partial[r * stride + c] over
// r in [0, nrows) for column c = 8 * blockIdx.x + (tid & 7).  256 threads = 8 columns x 32 row
// groups (a few hundred partial rows are typical: 32-way row parallelism keeps the load chain at ~16
// steps, with 4x more workgroups than a 32-column block); fixed summation order.
// Returns the sum in the threads with (tid >> 3) == 0.
constexpr int kFinalCols = 8;
__device__ __forceinline__ float block_colsum8(const float* __restrict__ partial, int nrows, int64_t stride, int c, bool valid,
                                               float (*red)[kFinalCols]) {
    const int cl = threadIdx.x & 7, rg = threadIdx.x >> 3;
    float s = 0.f;
    if (valid)
        for (int r = rg; r < nrows; r += 32) s += partial[(int64_t)r * stride + c];
    red[rg][cl] = s;
    __syncthreads();
    float t = 0.f;
    if (rg == 0) {
#pragma unroll
        for (int k = 0; k < 32; k += 4) t += (red[k][cl] + red[k + 1][cl]) + (red[k + 2][cl] + red[k + 3][cl]);
    }
    return t;
}

template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 x) { return bf2f(x); }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return f2bf(x); }

}  // namespace hvc
