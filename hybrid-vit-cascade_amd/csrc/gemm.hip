// Tiled MFMA GEMM with fused epilogues for gfx950:   z = drop(act(alpha * A B^T + bias));  C = residual + gate * z
//
// Covers every dense projection of the reference's ViT block and their gradients:
//   models/vit_components.py:26,28,74,75,77   qkv / q / kv / proj Linear
//   models/hybrid_vit_backbone.py:75-81       MLP Linear -> GELU(erf) -> Dropout -> Linear -> Dropout
//   models/hybrid_vit_backbone.py:123,128,139 residual adds (gated by AdaLN gate_sa / gate_mlp)
//   models/vit_components.py:131,144          AdaLN Linear(cond_dim, 6C)
//
// One kernel, four operand-layout variants:  C[i][j] = sum_k A(i,k) B(j,k)  where each operand is
// either k-contiguous in memory (fetched from LDS with ds_read_b128 row fragments) or k-major
// (its contraction index is the memory row index; fetched with ds_read_b64_tr_b16).  That gives
//   y  = x W^T          (A k-contig, B k-contig)     forward Linear, W is [out][in]
//   dx = dy W           (A k-contig, B k-major)      W read in place, no transposed copy
//   dW = dy^T x         (A k-major,  B k-major)      activations read in place
// Tile 128 x 128 x BK (BK = 64 for bf16 operands, 32 for fp32 operands carried as hi/lo bf16
// images), 4 waves in a 2 x 2 grid, each wave 64 x 64 = 2 x 2 MFMA 32x32x16 tiles, LDS double
// buffered with register-staged global loads issued one k-tile ahead.
#include "hvc_common.hip.h"
#include "hvc_kernels.h"
#include <stdlib.h>
#include <type_traits>

// Timing-only knock-outs for scripts/gemm_ko.py (wrong results by design; 0 in every shipped build)
#ifndef HVC_GEMM_KO
#define HVC_GEMM_KO 0
#endif

namespace hvc {
namespace {

// Workgroup tile = WM x WN wavefronts (4 in all), each computing MI x NI MFMA 32x32 blocks.
template <int WM_, int WN_, int MI_, int NI_>
struct TileCfg {
    static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_, BM = WM_ * MI_ * 32, BN = WN_ * NI_ * 32;
    static_assert(WM_ * WN_ == 4, "four wavefronts per workgroup");
};
using Tile128 = TileCfg<2, 2, 2, 2>;       // 128 x 128: every dense projection
using TileN32 = TileCfg<4, 1, 2, 1>;       // 256 x 32 : implicit-GEMM convolutions with <= 32 output channels
using TileN64 = TileCfg<4, 1, 2, 2>;       // 256 x 64 : ... with <= 64 output channels
using TileM32 = TileCfg<1, 4, 1, 1>;       // 32 x 128 : their weight gradients (rows = output channels)
using TileM64 = TileCfg<1, 4, 2, 1>;       // 64 x 128
using TileM32W = TileCfg<1, 4, 1, 2>;      // 32 x 256 : weight gradients of wide patch matrices - a decoded patch row serves four
using TileM64W = TileCfg<1, 4, 2, 2>;      // 64 x 256   column groups instead of two, twice the MFMAs per barrier


// n (<= 8) consecutive elements <-> 8 floats; vec: one 16-byte (bf16) / two 16-byte (fp32) accesses.
template <typename T>
__device__ __forceinline__ void load_n(const T* p, float (&v)[8], int n, bool vec) {
    if (vec) {
        Chunk8<T> c = load_chunk<T>(p, 8, true);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = chunk_get<T>(c, e);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < n ? to_f<T>(p[e]) : 0.f;
    }
}
template <typename T>
__device__ __forceinline__ void store_n(T* p, const float (&v)[8], int n, bool vec) {
    if (vec) {
        if constexpr (sizeof(T) == 2) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
            *reinterpret_cast<bf16x8*>(p) = o;
        } else {
            *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < n) p[e] = from_f<T>(v[e]);
    }
}

// Operand tile loader (EXT = tile extent along the operand's non-contracted index).
//                       KM = false: LDS image [EXT rows][BK] (row = i, k contiguous);
//                       KM = true : LDS image [BK rows][EXT] (row = k, i contiguous).
template <typename T, int BK, bool KM, int EXT = 128>
struct OperandTile {
    static constexpr int NS = NSplit<T>::value;
    static constexpr int CW = KM ? EXT : BK;
    static constexpr int ROWS = KM ? BK : EXT;
    static constexpr int CPR = CW / 8;
    static constexpr int NCH = ROWS * CPR;                 // 16-byte chunks per tile
    static constexpr int CPT = (NCH + 255) / 256;
    static constexpr bool FULL = NCH % 256 == 0;           // every thread owns CPT chunks (else: ids >= NCH are idle)
    static constexpr int IMG = ROWS * CW;
    Chunk8<T> reg[2][CPT];  // two staging sets: tiles are fetched two k-steps ahead of their use
    const T* next;          // this thread's first chunk in the next k-tile (interior fast path)
    int64_t rstep, kstep;   // elements between a thread's consecutive chunks / between consecutive k-tiles

    // i0: first row(i) of this output tile, k0: first k of the first k-tile that will be requested; tiles are then
    // requested in k order.
    __device__ __forceinline__ void init(const T* base, int64_t ld, int i0, int k0, int tid) {
        const int row = tid / CPR, ch = tid % CPR;
        rstep = (int64_t)(256 / CPR) * ld;
        if constexpr (!KM) { next = base + (int64_t)(i0 + row) * ld + k0 + ch * 8; kstep = BK; }
        else { next = base + (int64_t)(k0 + row) * ld + i0 + ch * 8; kstep = (int64_t)BK * ld; }
    }
    // base: operand pointer; ld; i0: first row(i) of this tile; ni: extent of i; k0: first k; nk: extent of k.
    // A tile that lies fully inside the operand (and is 16-byte addressable) takes unconditional vector loads from
    // incrementally advanced addresses; edge tiles go through the per-chunk bounds path.
    template <int SET>
    __device__ __forceinline__ void issue(const T* base, int64_t ld, int i0, int ni, int k0, int nk, bool vec, int tid) {
        if (FULL && vec && i0 + EXT <= ni && k0 + BK <= nk) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) reg[SET][c] = load_chunk<T>(next + c * rstep, 8, true);
        } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                int id = tid + 256 * c;
                if (!FULL && id >= NCH) break;
                int row = id / CPR, ch = id % CPR;
                if constexpr (!KM) {
                    int i = i0 + row, k = k0 + ch * 8;
                    int nv = (i < ni) ? (nk - k) : 0;
                    reg[SET][c] = load_chunk<T>(base + (int64_t)i * ld + k, nv > 8 ? 8 : nv, vec);
                } else {
                    int k = k0 + row, i = i0 + ch * 8;
                    int nv = (k < nk) ? (ni - i) : 0;
                    reg[SET][c] = load_chunk<T>(base + (int64_t)k * ld + i, nv > 8 ? 8 : nv, vec);
                }
            }
        }
        next += kstep;
    }
    // interior tile, decided once per workgroup: no condition at all around the loads (see the k loop)
    template <int SET>
    __device__ __forceinline__ void issue_fast() {
#if HVC_GEMM_KO != 2
#pragma unroll
        for (int c = 0; c < CPT; ++c) reg[SET][c] = load_chunk<T>(next + c * rstep, 8, true);
#endif
        next += kstep;
    }
    template <int SET>
    __device__ __forceinline__ void commit(bf16* images, int tid) {
        if (HVC_GEMM_KO == 4) return;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            int id = tid + 256 * c;
            if (!FULL && id >= NCH) break;
            int row = id / CPR, ch = id % CPR;
            bf16x8 im[NS];
            chunk_split<T>(reg[SET][c], im);
#pragma unroll
            for (int s = 0; s < NS; ++s) tile_store<CW>(images + s * IMG, row, ch, im[s]);
        }
    }
    // fragment for the 32 rows(i) starting at r0 and k-step ks (16 wide)
    static __device__ __forceinline__ bf16x8 frag(const bf16* image, int r0, int ks, int lane) {
        if constexpr (!KM) return row_frag<CW>(image, r0, 16 * ks, lane);
        else return tr_frag<CW, false>(image, 16 * ks, r0, lane);
    }
};

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.m) + n) >> f.l; }

// Operand tile whose elements are convolution patches gathered from a channels-last activation tensor (implicit GEMM:
// the patch matrix of spatial.hip's im2col never exists in HBM).  Same LDS image and fragments as OperandTile.
//   KM = false: tile rows = patch rows m (output positions), chunk = 8 channels of one tap.  A thread keeps the decoded
//               positions of its CPT rows for the whole k loop and decodes its (tap, channel) once per k-tile.
//   KM = true : tile rows = contraction index = patch rows m, columns = patch columns.  Eight consecutive lanes cover
//               128 bytes of one patch row; a thread owns NG column groups (their taps are fixed for the whole kernel)
//               of RPT patch rows, which it decodes once per k-tile.
template <typename T, int BK, bool KM, int EXT, bool BUF>
struct GatherTile : OperandTile<T, BK, KM, EXT> {
    using Base = OperandTile<T, BK, KM, EXT>;
    static constexpr int CPT = Base::CPT, CPR = Base::CPR;
    static_assert(Base::FULL, "gathered tiles keep every thread busy");
    static_assert(!KM || CPR % 8 == 0, "k-major gather: column groups of 8 chunks");
    static constexpr int NG = KM ? CPR / 8 : 1;            // KM = true: column groups per thread
    static constexpr int RPT = KM ? CPT / NG : CPT;        // rows per thread
    // BUF (activation tensor below 4 GiB, the usual case): the gather is BRANCH-FREE.  A chunk's byte offset is
    // (window origin of its patch row) + (offset of its tap) in wrapping 32-bit arithmetic - linear, so it is right
    // whenever the tap lies inside the volume, edge rows included - and an out-of-volume tap gets an offset beyond the
    // buffer, for which buffer_load returns zeros: no lane- or wave-level branch around the loads (branches split the
    // k loop into basic blocks, at whose joins hipcc waits for every load in flight - the prefetch turns synchronous).
    uint32_t org[RPT];                                     // KM = false: element offset of the row's window origin (mod 2^32)
    int pd[RPT], ph[RPT], pw[RPT], pb[RPT];                // KM = false: first source coordinate (o * stride - pad), b * SD
    int kd[NG], kh[NG], kw[NG], kc[NG];                    // KM = true : per-group tap and first channel (kd = -2^24: beyond K)
    uint32_t ktap[NG];                                     // KM = true : element offset of the group's tap inside a window
    uint32_t inside;                                       // KM = false: bit c = chunk row c is interior for every lane of the wave
    __amdgpu_buffer_rsrc_t rsrc;

    // KM = true: chunk c of thread tid sits at tile row (tid / 8) + 32 * (c / NG), chunk column (tid % 8) + 8 * (c % NG)
    static __device__ __forceinline__ int km_row(int tid, int c) { return (tid >> 3) + 32 * (c / NG); }
    static __device__ __forceinline__ int km_ch(int tid, int c) { return (tid & 7) + 8 * (c % NG); }

    static __device__ __forceinline__ void taps_of(const ConvGather& cg, int k, int& d, int& h, int& w, int& c) {
        const uint32_t tap = fdiv((uint32_t)k, cg.dC);
        c = k - (int)tap * cg.C;
        const uint32_t t2 = fdiv(tap, cg.dKW);
        w = (int)(tap - t2 * cg.KW);
        const uint32_t t3 = fdiv(t2, cg.dKH);
        h = (int)(t2 - t3 * cg.KH);
        d = (int)t3;
        if (cg.flip) { d = cg.KD - 1 - d; h = cg.KH - 1 - h; w = cg.KW - 1 - w; }
    }
    static __device__ __forceinline__ uint32_t tap_offset(const ConvGather& cg, int d, int h, int w, int c) {
        return (uint32_t)((d * cg.SH + h) * cg.SW + w) * (uint32_t)cg.C + (uint32_t)c;
    }
    static __device__ __forceinline__ void rows_of(const ConvGather& cg, int64_t m, int& d0, int& h0, int& w0, int& b0, uint32_t& origin) {
        const bool live = m < cg.M;
        const uint32_t mm = live ? (uint32_t)m : 0u;
        const uint32_t q1 = fdiv(mm, cg.dOW);
        const uint32_t ow = mm - q1 * cg.dOW.d;
        const uint32_t q2 = fdiv(q1, cg.dOH);
        const uint32_t oh = q1 - q2 * cg.dOH.d;
        const uint32_t b = fdiv(q2, cg.dOD);
        const uint32_t od = q2 - b * cg.dOD.d;
        d0 = live ? (int)od * cg.stride - cg.PD : -(1 << 24);     // a dead row fails every bounds test
        h0 = (int)oh * cg.stride - cg.PH;
        w0 = (int)ow * cg.stride - cg.PW;
        b0 = (int)b * cg.SD;
        origin = (uint32_t)(((b0 + d0) * cg.SH + h0) * cg.SW + w0) * (uint32_t)cg.C;
    }
    static __device__ __forceinline__ bool in_volume(const ConvGather& cg, int sd, int sh, int sw) {
        return (unsigned)sd < (unsigned)cg.SD && (unsigned)sh < (unsigned)cg.SH && (unsigned)sw < (unsigned)cg.SW;
    }
    // 64-bit fallback: lane-level bounds branch
    static __device__ __forceinline__ Chunk8<T> fetch(const ConvGather& cg, int b0, int sd, int sh, int sw, int c) {
        if (!in_volume(cg, sd, sh, sw)) return zero_chunk<T>();
        const int64_t pos = ((int64_t)(b0 + sd) * cg.SH + sh) * cg.SW + sw;
        return load_chunk<T>(reinterpret_cast<const T*>(cg.src) + pos * cg.C + c, 8, true);
    }
    __device__ __forceinline__ Chunk8<T> fetch_buf(const ConvGather& cg, bool ok, uint32_t elem_off) const {
        const uint32_t off = ok ? elem_off * (uint32_t)sizeof(T) : cg.bytes;      // = num_records: reads as zero
        Chunk8<T> r;
        if constexpr (sizeof(T) == 2) {
            r.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        } else {
            r.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            r.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 16, 0));
        }
        return r;
    }

    // x0: first tile row (KM = false: patch row i0;  KM = true: patch column j0)
    __device__ __forceinline__ void init(const ConvGather& cg, int x0, int tid) {
        if constexpr (BUF) rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(cg.src), 0, cg.bytes, 0x00020000);
        if constexpr (!KM) {
            inside = 0;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                rows_of(cg, (int64_t)x0 + (tid + 256 * c) / CPR, pd[c], ph[c], pw[c], pb[c], org[c]);
                // whole window inside the volume for every row of this wave's chunk row c: its bounds tests are skipped
                // (a wave-uniform branch around vector arithmetic only - the loads stay outside of it)
                const bool in = pd[c] >= 0 && ph[c] >= 0 && pw[c] >= 0 && pd[c] + cg.KD <= cg.SD && ph[c] + cg.KH <= cg.SH && pw[c] + cg.KW <= cg.SW;
                if (__all(in)) inside |= 1u << c;
            }
        } else {
#pragma unroll
            for (int gidx = 0; gidx < NG; ++gidx) {
                const int k = x0 + km_ch(tid, gidx) * 8;
                const bool live = k < cg.K;
                taps_of(cg, live ? k : 0, kd[gidx], kh[gidx], kw[gidx], kc[gidx]);
                ktap[gidx] = tap_offset(cg, kd[gidx], kh[gidx], kw[gidx], kc[gidx]);
                if (!live) kd[gidx] = -(1 << 24);
            }
        }
    }
    // k0: first contraction index of the tile (KM = false: patch column;  KM = true: patch row)
    template <int SET>
    __device__ __forceinline__ void issue(const ConvGather& cg, int64_t k0, int tid) {
        if constexpr (!KM) {
            const int k = (int)k0 + (tid % CPR) * 8;
            const bool live = k < cg.K;
            int d, h, w, cc;
            taps_of(cg, live ? k : 0, d, h, w, cc);
            if constexpr (BUF) {
                const uint32_t t = tap_offset(cg, d, h, w, cc);
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    bool ok = live;
                    if (!(inside & (1u << c))) ok = ok && in_volume(cg, pd[c] + d, ph[c] + h, pw[c] + w);
                    this->reg[SET][c] = fetch_buf(cg, ok, org[c] + t);
                }
            } else {
#pragma unroll
                for (int c = 0; c < CPT; ++c)
                    this->reg[SET][c] = live ? fetch(cg, pb[c], pd[c] + d, ph[c] + h, pw[c] + w, cc) : zero_chunk<T>();
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < RPT; ++rr) {
                int d0, h0, w0, b0;
                uint32_t origin;
                rows_of(cg, k0 + km_row(tid, rr * NG), d0, h0, w0, b0, origin);
#pragma unroll
                for (int gidx = 0; gidx < NG; ++gidx) {
                    if constexpr (BUF)
                        this->reg[SET][rr * NG + gidx] = fetch_buf(cg, in_volume(cg, d0 + kd[gidx], h0 + kh[gidx], w0 + kw[gidx]), origin + ktap[gidx]);
                    else
                        this->reg[SET][rr * NG + gidx] = fetch(cg, b0, d0 + kd[gidx], h0 + kh[gidx], w0 + kw[gidx], kc[gidx]);
                }
            }
        }
    }
    template <int SET>
    __device__ __forceinline__ void commit(bf16* images, int tid) {
        if constexpr (!KM) {
            Base::template commit<SET>(images, tid);
        } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                bf16x8 im[Base::NS];
                chunk_split<T>(this->reg[SET][c], im);
#pragma unroll
                for (int s = 0; s < Base::NS; ++s) tile_store<Base::CW>(images + s * Base::IMG, km_row(tid, c), km_ch(tid, c), im[s]);
            }
        }
    }
};

// FEAT >= 0 (persistent form only): the epilogue options are compile-time constants (kF* bits) instead of GemmArgs fields - the
// block's four fused projections get a straight-line epilogue (see `if constexpr (FEAT >= 0)` below).
constexpr int kFGelu = 1, kFGeluGrad = 2, kFZsave = 4, kFGate = 8, kFResidual = 16, kFDrop = 32;

template <typename TI, typename TO, bool AKM, bool BKM, int GATHER, typename Cfg, bool PERSIST = false, int FEAT = -1>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    static_assert(!PERSIST || GATHER == 0, "persistent form: both operands in memory");
    static_assert(FEAT < 0 || PERSIST, "static epilogues exist in the persistent form only");
    constexpr int NS = NSplit<TI>::value;
    constexpr int BK = sizeof(TI) == 2 ? 64 : 32;
    constexpr int kBM = Cfg::BM, kBN = Cfg::BN, MI = Cfg::MI, NI = Cfg::NI;
    constexpr bool GA = GATHER == 1 || GATHER == 3, GB = GATHER == 2 || GATHER == 4, GBUF = GATHER == 1 || GATHER == 2;
    static_assert(GATHER == 0 || (GA && !AKM && !BKM) || (GB && AKM && BKM), "gather variants: forward / dW layouts only");
    using TA = std::conditional_t<GA, GatherTile<TI, BK, false, kBM, GBUF>, OperandTile<TI, BK, AKM, kBM>>;
    using TB = std::conditional_t<GB, GatherTile<TI, BK, true, kBN, GBUF>, OperandTile<TI, BK, BKM, kBN>>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    constexpr int A_SZ = NS * TA::IMG, B_SZ = NS * TB::IMG;
    static_assert(2 * (A_SZ + B_SZ) * sizeof(bf16) >= kBM * kBN * sizeof(float), "epilogue staging reuses the operand buffers");
    auto At = [&](int buf) { return lds + buf * (A_SZ + B_SZ); };
    auto Bt = [&](int buf) { return lds + buf * (A_SZ + B_SZ) + A_SZ; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware work order.  Workgroup b runs on XCD b % 8 (round-robin dispatch), so XCD x is given the
    // contiguous chunk x of the work list and walks it in launch order.  The list is ordered so that neighbours
    // share operand panels through that XCD's L2: the tiles of one split-K slice are adjacent (they read the same
    // K-slice of both operands), and inside a slice the shorter tile dimension runs fastest, so a panel of the
    // LARGE operand (e.g. 128 activation rows x K) is used by all its column tiles back to back while the
    // small operand (weights) stays L2-resident anyway.
    const int tiles_m = (g.M + kBM - 1) / kBM, tiles_n = (g.N + kBN - 1) / kBN;
    const int nwg = tiles_m * tiles_n;
    // this workgroup's XCD chunk [c0, c0 + clen) of the work list and its position(s) in it: one position per workgroup
    // (grid = work list), or - persistent form - positions slot, slot + gridDim.x / 8, ... of a 512-workgroup grid
    int c0, clen;
    {
        const int total = nwg * g.splitk;
        const int q = total >> 3, rem = total & 7, xcd = blockIdx.x & 7;
        c0 = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        clen = q + (xcd < rem ? 1 : 0);
    }
    int slot = blockIdx.x >> 3;
    int split, i0, j0;                       // split-K slice (0 when splitk == 1) and tile origin of the current work item
    auto decode = [&](int w, int& sp, int& ti0, int& tj0) {
        sp = w / nwg;
        const int id = w % nwg;
        int tm, tn;
        if (tiles_m >= tiles_n) { tn = id % tiles_n; tm = id / tiles_n; }
        else { tm = id % tiles_m; tn = id / tiles_m; }
        ti0 = tm * kBM;
        tj0 = tn * kBN;
    };
    decode(c0 + slot, split, i0, j0);

    const TI* Ap = reinterpret_cast<const TI*>(g.A);
    const TI* Bp = reinterpret_cast<const TI*>(g.B);
    TA ta;
    TB tb;
    const int nkt_all = (g.K + BK - 1) / BK;
    const int per_split = (nkt_all + g.splitk - 1) / g.splitk;
    const int kt_begin = split * per_split;
    const int kt_end = min(nkt_all, kt_begin + per_split);
    // Software pipeline, two k-tiles deep: while tile kt is multiplied out of LDS buffer kt & 1, tile kt+1 sits in one
    // register set (written to the other LDS buffer at the end of the step) and tile kt+2 is in flight into the other
    // set - a load has two compute phases to arrive, which matters at the blocks' K = 256 (4 tiles).
    // FAST = the workgroup's tile is interior in both operands and every k-tile of its slice is full: the loads carry no
    // condition.  This matters beyond the saved compares: any branch around a load - even a uniform one - leaves hipcc unable
    // to count the loads in flight at the join, and it then waits for ALL of them (vmcnt(0)) before the LDS commit, i.e. also
    // for the tile requested in this very step; the two-deep prefetch degrades to one.  So the decision is taken once, outside
    // the k loop, and the bulk of the loop (every step that both requests and commits a tile) has no branch in it either.
    int li0 = i0, lj0 = j0;                  // origin of the tile whose k-tiles are being requested (the NEXT tile while a persistent
                                             // workgroup finishes the current one)
    auto issue_a = [&](auto set_tag, auto fast_tag, int kt) {
        constexpr int set = decltype(set_tag)::value;
        if constexpr (GA) ta.template issue<set>(g.cg, (int64_t)kt * BK, tid);
        else if constexpr (decltype(fast_tag)::value) ta.template issue_fast<set>();
        else ta.template issue<set>(Ap, g.lda, li0, g.M, kt * BK, g.K, g.vec_a != 0, tid);
    };
    auto issue_b = [&](auto set_tag, auto fast_tag, int kt) {
        constexpr int set = decltype(set_tag)::value;
        if constexpr (GB) tb.template issue<set>(g.cg, (int64_t)kt * BK, tid);
        else if constexpr (decltype(fast_tag)::value) tb.template issue_fast<set>();
        else tb.template issue<set>(Bp, g.ldb, lj0, g.N, kt * BK, g.K, g.vec_b != 0, tid);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    f32x16 acc[MI][NI];
    // first two k-tiles of the tile at (ti0, tj0) into the two register sets
    auto prologue = [&](auto fast_tag, int ti0, int tj0) {
        li0 = ti0;
        lj0 = tj0;
        if constexpr (GA) ta.init(g.cg, ti0, tid);
        else ta.init(Ap, g.lda, ti0, kt_begin * BK, tid);
        if constexpr (GB) tb.init(g.cg, tj0, tid);
        else tb.init(Bp, g.ldb, tj0, kt_begin * BK, tid);
        issue_a(S0{}, fast_tag, kt_begin);
        issue_b(S0{}, fast_tag, kt_begin);
    };
    auto prologue2 = [&](auto fast_tag) {   // second k-tile (the persistent form requests it after the previous tile's epilogue)
        if (kt_begin + 1 < kt_end) {
            issue_a(S1{}, fast_tag, kt_begin + 1);
            issue_b(S1{}, fast_tag, kt_begin + 1);
        }
    };
    auto mainloop = [&](auto fast_tag) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;
    ta.template commit<0>(At(0), tid);
    tb.template commit<0>(Bt(0), tid);
    __syncthreads();

    // one k-step: BUF = LDS buffer holding tile kt; register set BUF ^ 1 holds tile kt+1; tile kt+2 goes to set BUF.
    // BULK: kt + 2 < kt_end is known to the caller (request and commit unconditionally).
    auto kstep = [&](auto buf_tag, auto bulk_tag, int kt) {
        constexpr int buf = decltype(buf_tag)::value;
        constexpr bool BULK = decltype(bulk_tag)::value;
        if (BULK || kt + 2 < kt_end) {
            issue_a(buf_tag, fast_tag, kt + 2);
            issue_b(buf_tag, fast_tag, kt + 2);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[NS][MI], bfr[NS][NI];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[s][mi] = HVC_GEMM_KO == 5 ? bf16x8{} + (bf16)(float)(kt + mi) : TA::frag(At(buf) + s * TA::IMG, 32 * (MI * wm + mi), ks, lane);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bfr[s][ni] = HVC_GEMM_KO == 5 ? bf16x8{} + (bf16)(float)(kt + ni) : TB::frag(Bt(buf) + s * TB::IMG, 32 * (NI * wn + ni), ks, lane);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int sa = 0; sa < NS; ++sa)
#pragma unroll
                        for (int sb = 0; sb < NS; ++sb)
                            if (sa + sb <= 1) {
                                if (HVC_GEMM_KO == 3) acc[mi][ni][0] += (float)af[sa][mi][0] * (float)bfr[sb][ni][0];
                                else acc[mi][ni] = mfma32(af[sa][mi], bfr[sb][ni], acc[mi][ni]);
                            }
        }
        if (BULK || kt + 1 < kt_end) {
            ta.template commit<buf ^ 1>(At(buf ^ 1), tid);
            tb.template commit<buf ^ 1>(Bt(buf ^ 1), tid);
        }
        __syncthreads();
    };
    {
        int kt = kt_begin;
        for (; kt + 3 < kt_end; kt += 2) {         // both steps request a tile: no branch in this loop body
            kstep(S0{}, std::true_type{}, kt);
            kstep(S1{}, std::true_type{}, kt + 1);
        }
        for (; kt + 1 < kt_end; kt += 2) {
            kstep(S0{}, std::false_type{}, kt);
            kstep(S1{}, std::false_type{}, kt + 1);
        }
        if (kt < kt_end) kstep(S0{}, std::false_type{}, kt);
    }
    };
    auto epilogue = [&](int split, int i0, int j0, auto between) {
    if (HVC_GEMM_KO == 1) {
        float sum = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int x = 0; x < 16; ++x) sum += acc[mi][ni][x];
        if (sum == 1.2345e-30f) reinterpret_cast<float*>(g.C)[tid] = sum;
        between();
        return;
    }
    // ---- epilogue: stage the 128 x 128 fp32 tile through LDS (the operand buffers are free now) so that
    // every global access of the epilogue is a 16/32-byte row segment instead of a 2/4-byte column element.
    float* stage = reinterpret_cast<float*>(smem);          // [BM][BN] fp32 (128 x 128: 64 KiB = the operand double buffer)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int x = 0; x < 16; ++x)
                stage[(32 * (MI * wm + mi) + acc_row(x, h)) * kBN + 32 * (NI * wn + ni) + r] = acc[mi][ni][x];
    __syncthreads();

    constexpr int CHN = kBN / 8;             // 8-column chunks per tile row
    constexpr int RS = 256 / CHN;            // tile rows covered by one pass of the workgroup
    const int ch = tid % CHN;                // 8-column chunk of the tile row handled by this thread
    const int trow = tid / CHN;
    const int j = j0 + ch * 8;
    // (persistent form: interior tiles with 16-byte addressable epilogue operands only - none of the element-wise paths is compiled)
    const bool active = PERSIST || j < g.N;  // (threads past the last column still take part in `between`)
    const int nj = PERSIST ? 8 : active ? min(8, g.N - j) : 0;
    const bool vec = PERSIST || (g.vec_epi != 0 && nj == 8);

    if (g.splitk > 1) {   // raw fp32 partial tile -> slab `split` of the workspace; reduced by splitk_reduce_kernel
        between();
        if (!active) return;
        float* ws = g.workspace + (size_t)split * g.M * g.N;
        for (int row = trow; row < kBM; row += RS) {
            const int i = i0 + row;
            if (i >= g.M) break;
            const float* sp = stage + row * kBN + ch * 8;
            float* dst = ws + (size_t)i * g.N + j;
            if (vec) {
                *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(sp);
                *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(sp + 4);
            } else {
                for (int e = 0; e < nj; ++e) dst[e] = sp[e];
            }
        }
        return;
    }

    TO* Cp = reinterpret_cast<TO*>(g.C);
    if (g.epi_simple) {
        // C = alpha A B^T (+ bias): the projections' input gradients and the biased forward projections.  Straight-line: every
        // staged row of the thread is read first, then converted and stored - no branch and no wait between the stores (the generic
        // path below tests seven epilogue options per row, and hipcc cannot keep loads or stores in flight across their joins).
        constexpr int RPTS = kBM / RS;
        f32x4 a0[RPTS], a1[RPTS];
#pragma unroll
        for (int it = 0; it < RPTS; ++it) {
            const float* sp = stage + (trow + RS * it) * kBN + ch * 8;
            a0[it] = *reinterpret_cast<const f32x4*>(sp);
            a1[it] = *reinterpret_cast<const f32x4*>(sp + 4);
        }
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = 0.f;
        if (g.bias && active) {
            if (vec && (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + j), b1 = *reinterpret_cast<const f32x4*>(g.bias + j + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bs[e] = b0[e]; bs[4 + e] = b1[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < nj) bs[e] = g.bias[j + e];
            }
        }
        between();
        if (!active) return;
#pragma unroll
        for (int it = 0; it < RPTS; ++it) {
            const int i = i0 + trow + RS * it;
            if (!PERSIST && i >= g.M) break;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a0[it][e] * g.alpha + bs[e]; v[4 + e] = a1[it][e] * g.alpha + bs[4 + e]; }
            store_n<TO>(Cp + (int64_t)i * g.ldc + j, v, nj, vec);
        }
        return;
    }
    TO* auxp = reinterpret_cast<TO*>(g.aux);
    TI* zp = reinterpret_cast<TI*>(g.zsave);
    if constexpr (FEAT >= 0) {
        // Static form of the generic path below for interior, vector-addressable tiles whose 128 rows lie in one batch element:
        // every global read (bias, gate, residual rows / saved pre-activations) is issued first, then the next tile's operands,
        // then each row is read from the staged tile, finished and stored - no branch, counted waits only.
        constexpr bool F_GELU = (FEAT & kFGelu) != 0, F_GGRAD = (FEAT & kFGeluGrad) != 0, F_Z = (FEAT & kFZsave) != 0,
                       F_GATE = (FEAT & kFGate) != 0, F_RES = (FEAT & kFResidual) != 0, F_DROP = (FEAT & kFDrop) != 0;
        constexpr int R = kBM / RS;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, g0 = b0, g1 = b0;
        if (g.bias) {
            b0 = *reinterpret_cast<const f32x4*>(g.bias + j);
            b1 = *reinterpret_cast<const f32x4*>(g.bias + j + 4);
        }
        if constexpr (F_GATE) {
            const float* gp = g.gate + (int64_t)(i0 / g.rows_per_batch) * g.N + j;
            g0 = *reinterpret_cast<const f32x4*>(gp);
            g1 = *reinterpret_cast<const f32x4*>(gp + 4);
        }
        f32x4 r0[F_RES ? R : 1], r1[F_RES ? R : 1];
        Chunk8<TO> pre[F_GGRAD ? R : 1];
#pragma unroll
        for (int it = 0; it < R; ++it) {
            const int i = i0 + trow + RS * it;
            if constexpr (F_RES) {
                const float* rp = g.residual + (int64_t)(g.residual_rows > 0 ? i % g.residual_rows : i) * g.ldr + j;
                r0[it] = *reinterpret_cast<const f32x4*>(rp);
                r1[it] = *reinterpret_cast<const f32x4*>(rp + 4);
            }
            if constexpr (F_GGRAD) pre[it] = load_chunk<TO>(auxp + (int64_t)i * g.ldc + j, 8, true);
        }
        between();
        const uint32_t seed = F_DROP ? seed_with_counter(g.seed_lo, g.seed_ctr) : 0u;
#pragma unroll
        for (int it = 0; it < R; ++it) {
            const int row = trow + RS * it;
            const int i = i0 + row;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * kBN + ch * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * kBN + ch * 8 + 4);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a0[e] * g.alpha + b0[e]; v[4 + e] = a1[e] * g.alpha + b1[e]; }
            const int64_t co = (int64_t)i * g.ldc + j;
            if constexpr (F_GELU) {
                store_n<TO>(auxp + co, v, 8, true);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
            }
            if constexpr (F_GGRAD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(chunk_get<TO>(pre[it], e));
            }
            if constexpr (F_DROP) {
                bool keep[8];
                drop2d_keep8(drop2d_rowkey(seed, g.seed_hi, (uint64_t)i), (uint32_t)j, g.drop_thresh, keep);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = keep[e] ? v[e] * g.keep_scale : 0.f;
            }
            if constexpr (F_Z) store_n<TI>(zp + (int64_t)i * g.ldz + j, v, 8, true);
            if constexpr (F_GATE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] *= g0[e]; v[4 + e] *= g1[e]; }
            }
            if constexpr (F_RES) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[it][e]; v[4 + e] += r1[it][e]; }
            }
            store_n<TO>(Cp + co, v, 8, true);
        }
        return;
    }
    float bj[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bj[e] = 0.f;
    if constexpr (PERSIST) {
        if (g.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + j), b1 = *reinterpret_cast<const f32x4*>(g.bias + j + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bj[e] = b0[e]; bj[4 + e] = b1[e]; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bj[e] = (g.bias && e < nj) ? g.bias[j + e] : 0.f;
    }

    // Issue every global read of the epilogue (residual rows, saved pre-activations) before touching the staged tile,
    // so the 8 row segments of a thread cost one memory round trip instead of eight dependent ones.
    // (persistent form: in two batches of rows - the next tile's operand registers stay live through this epilogue)
    constexpr int RPT_ALL = kBM / RS;       // rows per thread
    constexpr int NBATCH = PERSIST && RPT_ALL >= 8 ? 4 : 1, RPT = RPT_ALL / NBATCH;
#pragma unroll
    for (int bt = 0; bt < NBATCH; ++bt) {
    const int trow_b = trow + RS * RPT * bt;
    float rres[RPT][8], rpre[RPT][8];
    if (g.residual) {
#pragma unroll
        for (int it = 0; it < RPT; ++it) {
            const int i = i0 + trow_b + RS * it;
            if (PERSIST || i < g.M) load_n<float>(g.residual + (int64_t)(g.residual_rows > 0 ? i % g.residual_rows : i) * g.ldr + j, rres[it], nj, vec);
        }
    }
    if (g.act == kActGeluGrad) {
#pragma unroll
        for (int it = 0; it < RPT; ++it) {
            const int i = i0 + trow_b + RS * it;
            if (PERSIST || i < g.M) load_n<TO>(auxp + (int64_t)i * g.ldc + j, rpre[it], nj, vec);
        }
    }
    if (bt == 0) between();                 // (persistent form: the next tile's first k-tiles are requested here, behind this tile's own reads)
    if (!active) return;
#pragma unroll
    for (int it = 0; it < RPT; ++it) {
        const int row = trow_b + RS * it;
        const int i = i0 + row;
        if (!PERSIST && i >= g.M) break;
        float v[8];
        {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * kBN + ch * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * kBN + ch * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a0[e] * g.alpha + bj[e]; v[4 + e] = a1[e] * g.alpha + bj[4 + e]; }
        }
        int64_t co = (int64_t)i * g.ldc + j;
        if (g.omap) {
            const uint32_t q1 = fdiv((uint32_t)i, g.oW), ow = (uint32_t)i - q1 * g.oW.d;
            const uint32_t q2 = fdiv(q1, g.oH), oh = q1 - q2 * g.oH.d;
            const uint32_t ob = fdiv(q2, g.oD), od = q2 - ob * g.oD.d;
            co = g.o_base + (int64_t)ob * g.o_sb + (int64_t)od * g.o_sd + (int64_t)oh * g.o_sh + (int64_t)ow * g.o_sw + j;
        }
        if (g.act == kActGelu) {
            if (auxp) store_n<TO>(auxp + co, v, nj, vec);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        } else if (g.act == kActGeluGrad) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(rpre[it][e]);
        }
        if (g.drop_thresh) {      // j is a multiple of 8: two 4-column hash groups per row segment
            bool keep[8];
            drop2d_keep8(drop2d_rowkey(seed_with_counter(g.seed_lo, g.seed_ctr), g.seed_hi, (uint64_t)i), (uint32_t)j, g.drop_thresh, keep);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = keep[e] ? v[e] * g.keep_scale : 0.f;
        }
        if (zp) store_n<TI>(zp + (int64_t)i * g.ldz + j, v, nj, vec);
        if (g.gate) {
            const float* gp = g.gate + (int64_t)(i / g.rows_per_batch) * g.N + j;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nj) v[e] *= gp[e];
        }
        if (g.residual) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rres[it][e];
        }
        store_n<TO>(Cp + co, v, nj, vec);
    }
    }
    };

    const bool k_full = (int64_t)kt_end * BK <= g.K;
    if constexpr (!PERSIST) {
        const bool all_fast = kt_begin < kt_end && k_full && (GA || (TA::FULL && g.vec_a != 0 && i0 + kBM <= g.M)) && (GB || (TB::FULL && g.vec_b != 0 && j0 + kBN <= g.N));
        if (all_fast) { prologue(std::true_type{}, i0, j0); prologue2(std::true_type{}); mainloop(std::true_type{}); }
        else { prologue(std::false_type{}, i0, j0); prologue2(std::false_type{}); mainloop(std::false_type{}); }
        epilogue(split, i0, j0, [] {});
        return;
    }
    // Persistent form (the launcher guarantees: no split-K, every tile interior, vector-addressable operands): a workgroup walks
    // positions slot, slot + step, ... of its XCD's chunk and requests the next tile's first two k-tiles from inside the current
    // tile's epilogue, so that their memory latency - half of a K = 256 tile's life otherwise - runs under the LDS staging and
    // the output stores.
    if constexpr (PERSIST) {
        const int step = gridDim.x >> 3;
        // Stagger: all 512 workgroups would otherwise walk load -> multiply -> store in step and the memory system would see a read
        // burst, silence, a write burst.  The second workgroup of every CU (the upper half of its XCD's slots: the dispatcher fills the XCD's
        // CUs once before it doubles up) starts late by g.persistent - 1 units of ~0.4 us, so that its stores fall under its
        // partner's loads; a persistent workgroup keeps that phase for its whole tile list.
        if (slot >= (step >> 1))
            for (int d = 1; d < g.persistent; ++d) __builtin_amdgcn_s_sleep(16);
        prologue(std::true_type{}, i0, j0);
        for (;;) {
            prologue2(std::true_type{});
            mainloop(std::true_type{});
            const int nslot = slot + step;
            if (nslot >= clen) break;
            int nsplit, ni0, nj0;
            decode(c0 + nslot, nsplit, ni0, nj0);
            epilogue(split, i0, j0, [&] { prologue(std::true_type{}, ni0, nj0); });
            __syncthreads();                 // the staged tile has been read: the operand buffers may be written again
            slot = nslot;
            i0 = ni0;
            j0 = nj0;
        }
        epilogue(split, i0, j0, [] {});
    }
}

template <typename TO, bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g) {
    const size_t total = (size_t)g.M * g.N;
    TO* Cp = reinterpret_cast<TO*>(g.C);
    if constexpr (VEC) {      // N % 4 == 0: one float4 per slab per thread, slabs summed in a fixed order (deterministic)
        const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
        if (e >= total) return;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 4 <= g.splitk; k += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(g.workspace + (size_t)k * total + e);
            const f32x4 b = *reinterpret_cast<const f32x4*>(g.workspace + (size_t)(k + 1) * total + e);
            const f32x4 c = *reinterpret_cast<const f32x4*>(g.workspace + (size_t)(k + 2) * total + e);
            const f32x4 d = *reinterpret_cast<const f32x4*>(g.workspace + (size_t)(k + 3) * total + e);
            s += a; s += b; s += c; s += d;
        }
        for (; k < g.splitk; ++k) s += *reinterpret_cast<const f32x4*>(g.workspace + (size_t)k * total + e);
        const size_t i = e / g.N, j = e % g.N;
        TO* o = Cp + i * g.ldc + j;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = from_f<TO>(s[t] * g.alpha);
    } else {
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            float s = 0.f;
            for (int k = 0; k < g.splitk; ++k) s += g.workspace[(size_t)k * total + e];
            const size_t i = e / g.N, j = e % g.N;
            Cp[i * g.ldc + j] = from_f<TO>(s * g.alpha);
        }
    }
}

// two resident workgroups per CU, a multiple of the eight XCDs (512 on MI355X)
inline int persistent_grid() { return (2 * cu_count()) & ~7; }

template <typename TI, typename TO, bool AKM, bool BKM, int GATHER = 0, typename Cfg = Tile128>
hipError_t launch(const GemmArgs& g_in, hipStream_t st) {
    GemmArgs g = g_in;
    constexpr int NS = NSplit<TI>::value;
    constexpr int BK = sizeof(TI) == 2 ? 64 : 32;
    constexpr int kBM = Cfg::BM, kBN = Cfg::BN;
    const size_t lds = (size_t)2 * NS * (kBM + kBN) * BK * sizeof(bf16);
    auto k = gemm_kernel<TI, TO, AKM, BKM, GATHER, Cfg>;
    static thread_local bool lds_raised = false;      // per instantiation; sticky attribute, set once (also keeps it out of graph captures)
    if (lds > 48 * 1024 && !lds_raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_raised = true;
    }
    const int tiles = ((g.M + kBM - 1) / kBM) * ((g.N + kBN - 1) / kBN);
    const int nkt = (g.K + BK - 1) / BK;
    // Split-K: a weight-gradient GEMM has a tiny output (N_out x K_in) and a contraction over all
    // tokens, i.e. a dozen tiles for 256 CUs.  Slice K over workgroups into fp32 slabs and reduce
    // them in a fixed order (deterministic; no float atomics).
    g.splitk = 1;
    const bool plain = !g.bias && g.act == kActNone && !g.aux && !g.zsave && !g.gate && !g.residual && !g.drop_thresh;
    if (plain && g.workspace && tiles <= 128 && nkt >= 8) {
        int want = 512 / tiles;                      // tiles * slices <= 512 = one resident round (2 workgroups on each of 256 CUs);
                                                     // rounding UP put e.g. 14 x 37 = 518 workgroups on the chip: two rounds, the second nearly empty
        if (want > nkt / 4) want = nkt / 4;          // >= 4 k-tiles per slice: slab traffic stays below the operand traffic
        const int64_t fit = g.workspace_floats / ((int64_t)g.M * g.N);
        if (want > fit) want = (int)fit;
        if (want >= 2) {
            const int per_split = (nkt + want - 1) / want;
            g.splitk = (nkt + per_split - 1) / per_split;      // no empty trailing slice (the kernel derives the same per-slice count)
        }
    }
    g.epi_simple = g.act == kActNone && !g.aux && !g.zsave && !g.gate && !g.residual && !g.drop_thresh && !g.omap;
    // Persistent form for the token-matrix projections (65536 x {256..1024} x {256..1024}: thousands of tiles of 4 - 16 k-steps):
    // every tile interior and vector-addressable, at least two tiles per resident workgroup.  HVC_GEMM_PERSISTENT=0 disables (A/B).
    const bool persistent_on = option(kOptGemmPersistent) != 0;      // read per launch: the tests run both forms in one process
    using TA = OperandTile<TI, BK, AKM, kBM>;
    using TB = OperandTile<TI, BK, BKM, kBN>;
    const int stagger = option(kOptGemmStagger) > 0 ? option(kOptGemmStagger) : 0;
    g.persistent = (1 + stagger) * (int)(persistent_on && GATHER == 0 && std::is_same_v<Cfg, Tile128> && g.splitk == 1 && tiles >= 1024 && TA::FULL && TB::FULL && g.vec_a && g.vec_b && g.vec_epi && !g.omap && (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0 &&
                   g.M % kBM == 0 && g.N % kBN == 0 && g.K % BK == 0);
    if constexpr (GATHER == 0 && std::is_same_v<Cfg, Tile128>) {      // (measured on the 128 x 128 tile only)
        if (g.persistent) {
            auto launch_p = [&](auto kp, bool& raised) -> hipError_t {
                if (lds > 48 * 1024 && !raised) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return e;
                    raised = true;
                }
                hipLaunchKernelGGL(kp, dim3(persistent_grid()), dim3(256), lds, st, g);
                return hipGetLastError();
            };
            // the block's fused projections (training form: dropout on) on 128 x 128 tiles: static epilogues
            if constexpr (std::is_same_v<Cfg, Tile128> && std::is_same_v<TI, bf16> && !AKM) {
                const int feat = (g.act == kActGelu ? kFGelu : 0) | (g.act == kActGeluGrad ? kFGeluGrad : 0) | (g.zsave ? kFZsave : 0) |
                                 (g.gate ? kFGate : 0) | (g.residual ? kFResidual : 0) | (g.drop_thresh ? kFDrop : 0);
                const bool aux_ok = g.act == kActNone ? !g.aux : g.aux != nullptr;
                const bool batch_ok = !g.gate || (g.rows_per_batch > 0 && g.rows_per_batch % kBM == 0 && (reinterpret_cast<uintptr_t>(g.gate) & 15) == 0);
                if (!g.epi_simple && aux_ok && batch_ok) {
                    static thread_local bool r1 = false, r2 = false, r3 = false, r4 = false;
                    if constexpr (!BKM && std::is_same_v<TO, bf16>)
                        if (feat == (kFGelu | kFDrop)) return launch_p(gemm_kernel<TI, TO, AKM, BKM, GATHER, Cfg, true, kFGelu | kFDrop>, r1);
                    if constexpr (BKM && std::is_same_v<TO, bf16>)
                        if (feat == (kFGeluGrad | kFDrop)) return launch_p(gemm_kernel<TI, TO, AKM, BKM, GATHER, Cfg, true, kFGeluGrad | kFDrop>, r2);
                    if constexpr (!BKM && std::is_same_v<TO, float>) {
                        if (feat == (kFZsave | kFGate | kFResidual | kFDrop))
                            return launch_p(gemm_kernel<TI, TO, AKM, BKM, GATHER, Cfg, true, kFZsave | kFGate | kFResidual | kFDrop>, r3);
                        if (feat == (kFResidual | kFDrop)) return launch_p(gemm_kernel<TI, TO, AKM, BKM, GATHER, Cfg, true, kFResidual | kFDrop>, r4);
                    }
                }
            }
            static thread_local bool lds_raised_p = false;
            return launch_p(gemm_kernel<TI, TO, AKM, BKM, GATHER, Cfg, true>, lds_raised_p);
        }
    }
    hipLaunchKernelGGL(k, dim3(tiles * g.splitk), dim3(256), lds, st, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || g.splitk == 1) return e;
    if (g.N % 4 == 0 && (reinterpret_cast<uintptr_t>(g.workspace) & 15) == 0) {
        const int blocks = (int)(((size_t)g.M * g.N / 4 + 255) / 256);
        hipLaunchKernelGGL((splitk_reduce_kernel<TO, true>), dim3(blocks), dim3(256), 0, st, g);
    } else {
        int blocks = (int)(((size_t)g.M * g.N + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL((splitk_reduce_kernel<TO, false>), dim3(blocks), dim3(256), 0, st, g);
    }
    return hipGetLastError();
}

// HVC_GEMM_HALF_TILE=0 pins the 128 x 128 tile (A/B switch for scripts/gemm_vs_blas.py)
inline bool use_half_tile(const GemmArgs& g) {
    const bool env_on = option(kOptGemmHalfTile) != 0;
    const int64_t tiles = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (!env_on || g.M <= 64) return false;
    if (gemm_workspace_floats(g.M, g.N, g.K) == 0) return tiles < 384;
    // split-K shapes with a handful of tiles (256 x 256 weight gradients: 4 tiles x 128 slices): twice the tiles means half the
    // slices for the same 512 workgroups, i.e. half the fp32 slab traffic
    return g.workspace != nullptr && tiles <= 4;              // measured: 256 x 256 -17 %, 12 and 16 tiles +10...15 %
}

template <typename TI, typename TO>
hipError_t launch_layout(const GemmArgs& g, hipStream_t st) {
    // Implicit-GEMM convolutions: tall tiles for few output channels (forward, dx), flat ones for their weight gradients, so
    // that a 32-channel layer does not multiply 96 columns of padding.
    if (g.gather == 1) {
        if (g.a_kmajor || g.b_kmajor) return hipErrorInvalidValue;
        if (g.N <= 32 && g.M >= 1024) return launch<TI, TO, false, false, 1, TileN32>(g, st);
        if (g.N <= 64 && g.M >= 1024) return launch<TI, TO, false, false, 1, TileN64>(g, st);
        return launch<TI, TO, false, false, 1>(g, st);
    }
    if (g.gather == 2) {
        if (!g.a_kmajor || !g.b_kmajor) return hipErrorInvalidValue;
        const bool wide = g.N >= 1024;
        if (g.M <= 32) return wide ? launch<TI, TO, true, true, 2, TileM32W>(g, st) : launch<TI, TO, true, true, 2, TileM32>(g, st);
        if (g.M <= 64) return wide ? launch<TI, TO, true, true, 2, TileM64W>(g, st) : launch<TI, TO, true, true, 2, TileM64>(g, st);
        return launch<TI, TO, true, true, 2>(g, st);
    }
    // activation tensors of 4 GiB and more: 64-bit addressed gather, 128 x 128 tiles only
    if (g.gather == 3) return (g.a_kmajor || g.b_kmajor) ? hipErrorInvalidValue : launch<TI, TO, false, false, 3>(g, st);
    if (g.gather == 4) return (!g.a_kmajor || !g.b_kmajor) ? hipErrorInvalidValue : launch<TI, TO, true, true, 4>(g, st);
    // Shapes that give the 512 workgroup slots of the chip fewer than ~3/4 of a round of 128 x 128 tiles (the 64^3 model's
    // token matrices: 16384 x 256 = 256 tiles) run on 64 x 128 tiles: twice the workgroups on the same latency chain.
    // Tall-skinny products (the single-channel convolution layers' patch GEMMs: 16.7 M rows x 32..64 columns at 256^3): 256-row
    // tiles without the padded columns - half the workgroups, each writing only what exists.
    if (!g.a_kmajor && g.N <= 64 && g.M >= 4096 && gemm_workspace_floats(g.M, g.N, g.K) == 0) {
        if (g.N <= 32) return g.b_kmajor ? launch<TI, TO, false, true, 0, TileN32>(g, st) : launch<TI, TO, false, false, 0, TileN32>(g, st);
        return g.b_kmajor ? launch<TI, TO, false, true, 0, TileN64>(g, st) : launch<TI, TO, false, false, 0, TileN64>(g, st);
    }
    if (use_half_tile(g)) {
        if (g.a_kmajor) return g.b_kmajor ? launch<TI, TO, true, true, 0, TileM64>(g, st) : launch<TI, TO, true, false, 0, TileM64>(g, st);
        return g.b_kmajor ? launch<TI, TO, false, true, 0, TileM64>(g, st) : launch<TI, TO, false, false, 0, TileM64>(g, st);
    }
    if (g.a_kmajor) return g.b_kmajor ? launch<TI, TO, true, true>(g, st) : launch<TI, TO, true, false>(g, st);
    return g.b_kmajor ? launch<TI, TO, false, true>(g, st) : launch<TI, TO, false, false>(g, st);
}

}  // namespace

int64_t gemm_workspace_floats(int M, int N, int K) {
    // 128 x 128 tiles; the flat weight-gradient tiles (M <= 64 rows, 128 columns) give the same tile count
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    if (tiles > 128 || K < 8 * 32) return 0;
    int want = (512 + tiles - 1) / tiles;
    const int nkt = (K + 31) / 32;
    if (want > nkt / 4) want = nkt / 4;
    return want >= 2 ? (int64_t)want * M * N : 0;
}

namespace {
// C[m][j] = alpha * sum_k A[m][k] B[j][k] + bias[j] for M <= 8 rows, fp32, both operands k-contiguous: the conditioning
// vectors' Linears (AdaLN 1024 -> 1536, time / context MLPs; M = batch).  A 128 x 128 MFMA tile would be 98 % padding
// and 12 workgroups; this streams the weight matrix once with one wavefront per output column (exact fp32 FMAs).
template <int MR>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const GemmArgs g) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= g.N) return;
    const float* A = reinterpret_cast<const float*>(g.A);
    const float* brow = reinterpret_cast<const float*>(g.B) + (int64_t)j * g.ldb;
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.f;
    for (int k = 4 * lane; k < g.K; k += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(brow + k);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < g.M) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(A + (int64_t)m * g.lda + k);
                acc[m] += (x[0] * w[0] + x[1] * w[1]) + (x[2] * w[2] + x[3] * w[3]);
            }
        }
    }
    float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float s = wave_sum(acc[m]);
        if (lane == 0 && m < g.M) C[(int64_t)m * g.ldc + j] = s * g.alpha + (g.bias ? g.bias[j] : 0.f);
    }
}
}  // namespace

hipError_t gemm_launch(const GemmArgs& g, hipStream_t st) {
    if (g.in_bf16) return g.out_bf16 ? launch_layout<bf16, bf16>(g, st) : launch_layout<bf16, float>(g, st);
    if (g.out_bf16) return hipErrorInvalidValue;
    const bool plain_rows = g.M <= 8 && !g.gather && !g.a_kmajor && !g.b_kmajor && g.act == kActNone && !g.aux && !g.zsave && !g.gate && !g.residual &&
                            !g.drop_thresh && (g.K % 4 == 0) && (g.lda % 4 == 0) && (g.ldb % 4 == 0) &&
                            (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0;
    if (plain_rows) {
        if (g.M <= 2) hipLaunchKernelGGL(gemv_rows_kernel<2>, dim3((g.N + 3) / 4), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(gemv_rows_kernel<8>, dim3((g.N + 3) / 4), dim3(256), 0, st, g);
        return hipGetLastError();
    }
    return launch_layout<float, float>(g, st);
}

}  // namespace hvc
