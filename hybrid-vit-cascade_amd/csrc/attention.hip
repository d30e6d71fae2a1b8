// Fused multi-head attention for gfx950 (MI355X): forward, and a two-kernel backward.
//
// Replaces the reference's materialised-score attention
//   models/vit_components.py:41-51   (self-attention : qkv split, q@k^T*scale, softmax, dropout, @v)
//   models/vit_components.py:95-113  (cross-attention: q / kv split, same core)
// with flash-style kernels that never write the (B,h,N,N) score matrix to HBM.
//
// Layout: q/k/v/o/do are addressed as  ptr[b*sb + n*sn + h*sh + d]  (d contiguous), so the
// kernels read q,k,v straight out of the packed qkv / kv projection outputs and write o in
// (B,N,h*d) order, which is what the reference's transpose(1,2).reshape produces.
//
// Work decomposition (all kernels: 256 threads = 4 waves, <= 256 registers so that 2-3 workgroups
// share a CU and one wave's softmax VALU work overlaps another wave's MFMAs):
//   fwd    : workgroup = 128 query rows of one (b,h); wave = 32 rows; loop over 64-key tiles
//            staged in LDS (double buffered, register-staged global loads issued one tile ahead).
//            S^T = K Q^T (key rows in registers, query on the lane) so max / sum / rescale are
//            per-lane scalars; P feeds the PV product straight from the accumulators
//            (accumulator-as-B-operand) and V is consumed through ds_read_b64_tr_b16.
//            The running max is only raised (and O, l rescaled) when some row's tile max exceeds it
//            by more than 2^kRescaleLog2 -- a wave-uniform, rarely taken branch.
//   bwd dQ : same decomposition; recomputes S^T and dP^T = V dO^T, dQ += dS K (K via tr read).
//   bwd dKV: workgroup = 128 keys of one (b,h); wave = 32 keys held in registers; loop over
//            64-query tiles (Q, dO in LDS, read row-wise for S / dP and transposed for dV / dK).
// The split backward recomputes S twice (7 products instead of 5) in exchange for having no
// cross-workgroup reduction: dQ, dK, dV are bitwise reproducible.
#include <cstdlib>
#include <type_traits>

#include "hvc_common.hip.h"
#include "hvc_kernels.h"
#include "attn_dropout.hip.h"

namespace hvc {

namespace {

constexpr int kKT = 64;     // keys (or queries, in dKV) per LDS tile
constexpr int kQB = 128;    // rows per workgroup
// dK/dV kernel's dropout-lot tile: one row per query of the tile, four 32-key parts (one per wavefront) of 64 bytes each, stored
// 80 bytes apart, rows 320 bytes apart.  The generator's ds_write_b128 is served in groups of 8 lanes = 2 query rows x 4 parts:
// part starts of 0 / 80 / 160 / 240 bytes and a row step of 320 bytes (= 16 banks mod 32) put the eight 16-byte stores of a group
// on eight distinct 4-bank groups.  With the parts contiguous (64-byte steps) parts 0 / 2 and 1 / 3 met on the same banks:
// rocprofv3 counted SQ_LDS_BANK_CONFLICT = 1 cycle per MFMA (134 M per launch at N = 32768, profiles/r01_pmc_attention_*) for this
// kernel, all of it from these stores (the fragment reads of the Q / dO tiles are conflict-free, as in the forward and dQ kernels).
// (That is the eight-wavefront tile.  The four-wavefront tile keeps its parts contiguous and pads the ROW instead - 4 x 64 + 16 bytes: two
// rows x four parts of a store group fall on banks 0 / 16 / 32 / 48 and 4 / 20 / 36 / 52, the next rows on + 8, + 12 - which takes the tile from
// 40 to 34 KB: at d = 32 a workgroup then needs 51 KB and THREE of them share a CU, three wavefronts per SIMD: dK/dV -5 % without dropout, round 4.)
template <int WAVES> struct LotTile {
    static constexpr int PART = WAVES == 4 ? 32 : 32 + 8;            // 16-bit lots from one part's start to the next
    static constexpr int ROW = WAVES == 4 ? 4 * 32 + 8 : 8 * (32 + 8);  // lots per tile row
};
constexpr float kRescaleLog2 = 6.f;   // deferred running-max update: P stays <= 2^6 between rescales
constexpr float kRescaleSum = 1024.f; // 64-row forward: the same test on a half-row sum of 16 probabilities (16 * 2^6)

// 8 elements of row n (zeros when n >= nrows).  VEC: one clamped 16-byte load + select (no branch).
template <typename T, bool VEC>
__device__ __forceinline__ Chunk8<T> load_row_chunk(const T* base, int64_t sn, int n, int nrows, int col) {
    if constexpr (VEC) {
        const int nc = n < nrows ? n : nrows - 1;
        Chunk8<T> c = load_chunk<T>(base + (int64_t)nc * sn + col, 8, true);
        if (n >= nrows) c = zero_chunk<T>();
        return c;
    } else {
        return load_chunk<T>(base + (int64_t)n * sn + col, n < nrows ? 8 : 0, false);
    }
}

// Tile staging, two forms.
//  * register staged (fp32 operands, unaligned / ragged-stride operands): global loads into registers one tile ahead, ds_write_b128
//    into the swizzled image at the end of the tile.
//  * LDS-DMA (bf16 operands with 16-byte addressable rows; HVC_ATTN_DMA=0 at build time pins the register form): the tile goes
//    from global memory straight into the LDS image with global_load_lds_dwordx4 - no staging registers, no ds_write_b128 (whose
//    13 cycles each on the LDS store path were, with the loads, 8 - 14 % of the dK/dV kernel: profiles/r03_attention_*).  One
//    wave-instruction writes 64 lanes x 16 bytes CONTIGUOUSLY (1 KiB = 8 or 16 tile rows), so the image's XOR swizzle moves to
//    the source side: the lane that fills physical slot s of row r fetches logical chunk s ^ sw(r) of that row (an involution;
//    the eight / four lanes of a row still cover its whole 128 / 64 bytes, so global coalescing is unchanged).  Rows past the end
//    of the operand are clamped to its last row (finite values; their probabilities are zero either way).  Nothing tracks a
//    DMA's arrival but the issuing wave's vmcnt: wait() before the barrier that ends the tile.
#ifndef HVC_ATTN_DMA
#define HVC_ATTN_DMA 1
#endif
// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to the 1 KiB at the wave-uniform LDS address dst.
// Inline asm on purpose: issued through __builtin_amdgcn_global_load_lds, hipcc puts an s_waitcnt vmcnt(0) in front of the next
// LDS read that it cannot prove disjoint from the piece - i.e. in the middle of the tile, where it exposes the whole load latency.
// As an asm statement the piece is invisible to that bookkeeping (its arrival is ours to wait for: TileLoader::wait()); the
// compiler's own vmcnt(N) waits stay correct, they can only wait for more than they meant to.  M0 (the LDS base of the piece) is
// compiler-reserved: saved, written and restored inside the one statement, with the wait state its reader needs.
__device__ __forceinline__ void lds_dma16(const void* src, const bf16* dst) {
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst);      // generic -> LDS offset: the low word
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}
// The same piece with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit vector address arithmetic.
__device__ __forceinline__ void lds_dma16_s(const void* sbase, uint32_t voff, const bf16* dst) {
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <typename T, int D, bool VEC>
struct TileLoader {
    static constexpr int NS = NSplit<T>::value;
    static constexpr int CPR = D / 8;                  // 16-byte chunks per row
    static constexpr int CHUNKS = kKT * CPR;           // per tile image
    static constexpr int CPT = CHUNKS / 256;           // chunks per thread
    static constexpr bool DMA = HVC_ATTN_DMA && VEC && sizeof(T) == 2;
    Chunk8<T> reg[DMA ? 1 : CPT];
    const T* next;                                     // this thread's first chunk in the next tile (VEC path)
    int64_t step, rstep;                               // elements between consecutive tiles / between a thread's chunks
    int drow, dch;                                     // DMA: tile row / logical chunk of this lane's first piece
    int dwave;                                         // DMA: element offset of this wavefront's share of a piece inside the image

    // row0 = first row of the first tile this loader will be asked for; tiles are then requested in order.
    __device__ __forceinline__ void init(const T* base, int64_t sn, int row0, int tid) {
        step = (int64_t)kKT * sn;
        rstep = (int64_t)(256 / CPR) * sn;             // chunk c of a thread sits 256 / CPR rows below chunk c - 1
        if constexpr (DMA) {
            drow = tid / CPR;                          // physical chunk tid (+ 256 per piece) = row tid / CPR, slot tid % CPR
            dch = (tile_off<D>(drow, tid % CPR) - drow * D) >> 3;      // the logical chunk stored in that slot (XOR: involution);
            dwave = __builtin_amdgcn_readfirstlane((tid >> 6) * 64 * 8);   // rows 256 / CPR further down swizzle the same way
            next = base + (int64_t)(row0 + drow) * sn + dch * 8;
        } else {
            next = base + (int64_t)(row0 + tid / CPR) * sn + (tid % CPR) * 8;
        }
    }
    // Full tiles take the incremental addresses (one 64-bit add per chunk, no bounds selects); a ragged or
    // unaligned tile goes through the clamped / per-element path.  dst: the image(s) of the tile being requested (DMA form).
    __device__ __forceinline__ void issue(const T* base, int64_t sn, int row0, int nrows, int tid, bf16* dst) {
        if constexpr (DMA) {
            const bool full = row0 + kKT <= nrows;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const T* src = full ? next + i * rstep
                                    : base + (int64_t)min(row0 + drow + i * (256 / CPR), nrows - 1) * sn + dch * 8;
                lds_dma16(src, dst + (i * 256 * 8 + dwave));
            }
        } else if (VEC && row0 + kKT <= nrows) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) reg[i] = load_chunk<T>(next + i * rstep, 8, true);
        } else {
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                int c = tid + 256 * i;
                int row = c / CPR, ch = c % CPR;
                reg[i] = load_row_chunk<T, VEC>(base, sn, row0 + row, nrows, ch * 8);
            }
        }
        next += step;
    }
    // images: NS consecutive tiles of kKT*D bf16
    __device__ __forceinline__ void commit(bf16* images, int tid) {
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                int c = tid + 256 * i;
                int row = c / CPR, ch = c % CPR;
                bf16x8 im[NS];
                chunk_split<T>(reg[i], im);
#pragma unroll
                for (int s = 0; s < NS; ++s) tile_store<D>(images + s * (kKT * D), row, ch, im[s]);
            }
        }
    }
    // every DMA this wavefront issued has landed (call before the barrier that publishes the tile)
    static __device__ __forceinline__ void wait() {
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
};

__device__ __forceinline__ void block_map(int id, int nbh, int nblk, int& bh, int& blk) {
    // Workgroups b and b+8 share an XCD (round-robin dispatch): keep all blocks of one (b,h)
    // on one XCD so its K/V (or Q/dO) stream is served by that XCD's L2.  Speed only.
    if ((nbh & 7) == 0) {
        int xcd = id & 7, slot = id >> 3;
        bh = xcd + 8 * (slot / nblk);
        blk = slot % nblk;
    } else {
        bh = id / nblk;
        blk = id % nblk;
    }
}

template <typename T, int NS, int DS, bool VEC>
__device__ __forceinline__ void load_row_frags(const T* rowp, int h, bool valid, bf16x8 (&f)[NS][DS]) {
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        Chunk8<T> c = VEC ? load_chunk<T>(rowp + 16 * s + 8 * h, 8, true) : load_chunk<T>(rowp + 16 * s + 8 * h, 8, false);
        if (!valid) c = zero_chunk<T>();
        bf16x8 im[NS];
        chunk_split<T>(c, im);
#pragma unroll
        for (int x = 0; x < NS; ++x) f[x][s] = im[x];
    }
}

// Row fragments of c * row (the softmax scale * log2(e) rides on the register-resident operand, so the score
// tiles leave the MFMA chain in exp2 units).
template <typename T, int NS, int DS, bool VEC>
__device__ __forceinline__ void load_row_frags_scaled(const T* rowp, int h, bool valid, float c, bf16x8 (&f)[NS][DS]) {
#pragma unroll
    for (int s = 0; s < DS; ++s) {
        Chunk8<T> ch = VEC ? load_chunk<T>(rowp + 16 * s + 8 * h, 8, true) : load_chunk<T>(rowp + 16 * s + 8 * h, 8, false);
        if (!valid) ch = zero_chunk<T>();
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = chunk_get<T>(ch, j) * c;
        bf16x8 im[NS];
        acc_split<NS>(x, im);
#pragma unroll
        for (int t = 0; t < NS; ++t) f[t][s] = im[t];
    }
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <typename T, int D, bool DROP, bool VEC>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 3 : 2) void attn_fwd_kernel(const AttnArgs a_in) {
    AttnArgs a = a_in;
    a.seed_lo = seed_with_counter(a_in.seed_lo, a_in.seed_ctr);
    constexpr int NS = NSplit<T>::value;
    constexpr int TILE = kKT * D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);   // [buf][K|V][NS][TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int nqb = (a.Nq + kQB - 1) / kQB;
    int bh, qb;
    block_map(blockIdx.x, a.B * a.H, nqb, bh, qb);
    const int b = bh / a.H, hh = bh % a.H;
    const T* qp = reinterpret_cast<const T*>(a.q) + b * a.q_sb + hh * a.q_sh;
    const T* kp = reinterpret_cast<const T*>(a.k) + b * a.k_sb + hh * a.k_sh;
    const T* vp = reinterpret_cast<const T*>(a.v) + b * a.v_sb + hh * a.v_sh;
    T* op = reinterpret_cast<T*>(a.o) + b * a.o_sb + hh * a.o_sh;

    const int q0 = qb * kQB + wave * 32;
    const int qrow = q0 + r;
    const bool qvalid = qrow < a.Nq;
    const int qrow_c = qvalid ? qrow : a.Nq - 1;

    // Q rows carry scale * log2(e): S^T = K (c Q)^T is already in exp2 units.
    const float sl2 = a.scale * kLog2e;
    bf16x8 qf[NS][D / 16];
    load_row_frags_scaled<T, NS, D / 16, VEC>(qp + (int64_t)qrow_c * a.q_sn, h, true, sl2, qf);

    TileLoader<T, D, VEC> kl, vl;
    auto Kt = [&](int buf) { return lds + (buf * 2 + 0) * NS * TILE; };
    auto Vt = [&](int buf) { return lds + (buf * 2 + 1) * NS * TILE; };

    const int nt = (a.Nk + kKT - 1) / kKT;
    kl.init(kp, a.k_sn, 0, tid);
    vl.init(vp, a.v_sn, 0, tid);
    kl.issue(kp, a.k_sn, 0, a.Nk, tid, Kt(0));
    vl.issue(vp, a.v_sn, 0, a.Nk, tid, Vt(0));
    kl.commit(Kt(0), tid);
    vl.commit(Vt(0), tid);
    kl.wait();
    __syncthreads();

    // Reference exponent msc (exp2 units) of this lane's query row; p = exp2(s - msc) <= 2^kRescaleLog2.  -msc is
    // kept replicated in a 16-register block that seeds the S^T accumulators, so the MFMA chain itself delivers
    // s - msc and the exponentials need no per-element subtraction.
    float msc = 0.f;
    f32x16 negm;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;
    float l = 0.f;
    f32x16 o[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;

    const uint32_t rowkey = DROP ? drop_rowkey(a, bh, qrow_c) ^ (h ? kGrpH : 0u) : 0u;      // lane half = low bit of the group index
    const uint32_t tm1x2 = DROP ? ((uint32_t)(drop_ts(a) - 1) & 0xffffu) * 0x10001u : 0u;

    // Per-lane LDS addresses of the K row fragments (one per 16-wide d step) and the transposed V fragments (two row
    // groups per 32-wide d block); tile buffer, key sub-tile and split image enter as immediate offsets.
    const bf16* kaddr[D / 16];
    const bf16* vaddr[D / 32][2];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) kaddr[s] = lds + row_frag_lane_off<D>(16 * s, lane);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
        int oa, ob;
        tr_frag_lane_off<D, true>(32 * dt, lane, oa, ob);
        vaddr[dt][0] = lds + oa;
        vaddr[dt][1] = lds + ob;
    }

    // One 64-key tile.  MASK is only instantiated for a ragged last tile, so the full tiles carry no
    // per-element bounds selects; the LDS buffer index is a compile-time constant (the tile loop is unrolled by two).
    auto tile = [&](auto mask_tag, auto buf_tag, int t) {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr int buf = decltype(buf_tag)::value;
        constexpr int KOFF = (buf * 2 + 0) * NS * TILE, VOFF = (buf * 2 + 1) * NS * TILE;
        if (t + 1 < nt) {
            kl.issue(kp, a.k_sn, (t + 1) * kKT, a.Nk, tid, Kt(buf ^ 1));
            vl.issue(vp, a.v_sn, (t + 1) * kKT, a.Nk, tid, Vt(buf ^ 1));
        }
        // S^T[key][q] - msc[q] = K (c Q)^T + (-msc)
        f32x16 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[kt] = negm;
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
#pragma unroll
                for (int sa = 0; sa < NS; ++sa) {
                    bf16x8 kf = *reinterpret_cast<const bf16x8*>(kaddr[s] + (KOFF + sa * TILE + 32 * kt * D));
#pragma unroll
                    for (int sb = 0; sb < NS; ++sb)
                        if (sa + sb <= 1) st[kt] = mfma32(kf, qf[sb][s], st[kt]);
                }
            }
        }
        const int kbase = t * kKT;
        const uint32_t rk_tile = rowkey + (uint32_t)t * kTileAdd;
        if constexpr (MASK) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kbase + 32 * kt + acc_row(i, h) >= a.Nk) st[kt][i] = -INFINITY;
        }
        float mloc = -INFINITY;      // row maximum relative to the reference
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, st[kt][i]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        // Deferred rescale: move the reference only when some row would overflow the 2^kRescaleLog2 head-room,
        // and on the first tile, where every row adopts its own maximum.  Wave-uniform branch; when taken, O, l,
        // the reference block and this tile's scores move together, exactly once, before P is formed.
        if (t == 0 || __any(mloc > kRescaleLog2)) {
            const float shift = t == 0 ? mloc : fmaxf(mloc, 0.f);
            const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-shift);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            msc += shift;
#pragma unroll
            for (int i = 0; i < 16; ++i) negm[i] = -msc;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) st[kt][i] -= shift;
        }
        float rs = 0.f;      // plain adds: the packed form (v_pk_add_f32) measured 2-6 % slower here, its dependent chain needs wait states
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(st[kt][i]);
                rs += p;
                st[kt][i] = p;
            }
        l += rs;
        // O^T[d][q] += V^T P^T
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = st[kt][8 * s2 + j];
                bf16x8 pf[NS];
                acc_split<NS>(x, pf);
                if constexpr (DROP) {      // packed keep masks onto the bf16 pairs: registers 8 s2 .. + 7 = two 4-key groups
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {      // group j = 8 kt + 2 (2 s2 + g2) + h of tile t
                        const uint32_t m = rk_tile ^ (drop_grp_a(kt) ^ drop_grp_b(2 * s2 + g2));
                        uint32_t la, lb;
                        drop_lots4(m, la, lb);
                        const uint32_t ma = drop_keepmask2(la, tm1x2), mb = drop_keepmask2(lb, tm1x2);
#pragma unroll
                        for (int sa = 0; sa < NS; ++sa) {
                            u32x4 w = __builtin_bit_cast(u32x4, pf[sa]);
                            w[2 * g2] &= ma;
                            w[2 * g2 + 1] &= mb;
                            pf[sa] = __builtin_bit_cast(bf16x8, w);
                        }
                    }
                }
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
                    for (int sa = 0; sa < NS; ++sa) {
                        const int VO = VOFF + (32 * kt + 16 * s2) * D;
                        bf16x8 vf = tr_frag_at(vaddr[dt][0] + (VO + sa * TILE), vaddr[dt][1] + (VO + sa * TILE));
#pragma unroll
                        for (int sb = 0; sb < NS; ++sb)
                            if (sa + sb <= 1) o[dt] = mfma32(vf, pf[sb], o[dt]);
                    }
                }
            }
        }
        if (t + 1 < nt) {
            kl.commit(Kt(buf ^ 1), tid);
            vl.commit(Vt(buf ^ 1), tid);
        }
        kl.wait();
        __syncthreads();
    };
    const bool ragged = (a.Nk % kKT) != 0;
    const int nfull = ragged ? nt - 1 : nt;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 1 < nfull; t += 2) {
        tile(std::false_type{}, B0{}, t);
        tile(std::false_type{}, B1{}, t + 1);
    }
    if (t < nfull) tile(std::false_type{}, B0{}, t);            // t is even here
    if (ragged) {
        if ((nt - 1) & 1) tile(std::true_type{}, B1{}, nt - 1);
        else tile(std::true_type{}, B0{}, nt - 1);
    }

    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = (DROP ? a.keep_scale : 1.f) / ltot;
    if (qvalid) {
        T* orow = op + (int64_t)qrow * a.o_sn;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * dt + 8 * g + 4 * h;
                if constexpr (sizeof(T) == 2) {
                    bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = f2bf(o[dt][4 * g + j] * inv);
                    if constexpr (VEC) *reinterpret_cast<bf16x4*>(orow + d0) = w;
                    else { for (int j = 0; j < 4; ++j) orow[d0 + j] = w[j]; }
                } else {
                    f32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = o[dt][4 * g + j] * inv;
                    if constexpr (VEC) *reinterpret_cast<f32x4*>(orow + d0) = w;
                    else { for (int j = 0; j < 4; ++j) orow[d0 + j] = w[j]; }
                }
            }
        if (h == 0) a.lse[(int64_t)bh * a.Nq + qrow] = (msc + __builtin_amdgcn_logf(ltot)) * kLn2;
    }
}

// ---------------------------------------------------------------------------------
// forward, 64 query rows per wavefront (bf16, 16-byte addressable operands)
// ---------------------------------------------------------------------------------
// Same decomposition as attn_fwd_kernel with two 32-row query blocks per wavefront and 32-key LDS tiles: every K row
// fragment and transposed V fragment read from LDS feeds two MFMAs (one per query block), and a workgroup's K/V tile
// traffic (global loads, LDS writes, barrier) is shared by 256 query rows instead of 128.  Per 16 MFMAs: 12 LDS
// fragment reads instead of 24, 2 tile-chunk loads per thread instead of 4.  Dropout lots are the same function of
// (row, key) as everywhere else: a 32-key tile t is half t & 1 of the 64-key hash tile t >> 1.
constexpr int kQB2 = 256, kKT2 = 32;

// WAVES = 8 (d = 64, chosen by launch_fwd): 512 query rows per workgroup, one workgroup per CU; threads 0-255 load the K chunk of a
// tile, threads 256-511 its V chunk.
template <int D, bool DROP, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void attn_fwd2_kernel(const AttnArgs a_in) {
    constexpr int QBW = 64 * WAVES, THREADS = 64 * WAVES;
    AttnArgs a = a_in;
    a.seed_lo = seed_with_counter(a_in.seed_lo, a_in.seed_ctr);
    using T = bf16;
    constexpr int TILE = kKT2 * D;
    constexpr int CPR = D / 8;                         // 16-byte chunks per tile row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);         // [buf][K|V][TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int nqb = (a.Nq + QBW - 1) / QBW;
    int bh, qb;
    block_map(blockIdx.x, a.B * a.H, nqb, bh, qb);
    const int b = bh / a.H, hh = bh % a.H;
    const T* qp = reinterpret_cast<const T*>(a.q) + b * a.q_sb + hh * a.q_sh;
    const T* kp = reinterpret_cast<const T*>(a.k) + b * a.k_sb + hh * a.k_sh;
    const T* vp = reinterpret_cast<const T*>(a.v) + b * a.v_sb + hh * a.v_sh;
    T* op = reinterpret_cast<T*>(a.o) + b * a.o_sb + hh * a.o_sh;

    const int q0 = qb * QBW + wave * 64;
    const float sl2 = a.scale * kLog2e;
    bf16x8 qf[2][1][D / 16];
    uint32_t rowkey[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int qc = min(q0 + 32 * blk + r, a.Nq - 1);
        load_row_frags_scaled<T, 1, D / 16, true>(qp + (int64_t)qc * a.q_sn, h, true, sl2, qf[blk]);
        rowkey[blk] = DROP ? drop_rowkey(a, bh, qc) ^ (h ? kGrpH : 0u) : 0u;
    }
    const uint32_t tm1x2 = DROP ? ((uint32_t)(drop_ts(a) - 1) & 0xffffu) * 0x10001u : 0u;

    // tile loader: a 32-key tile is 32 * D / 8 chunks of 16 bytes per operand.  D = 64: 256 chunks, every thread loads one of K
    // and one of V.  D = 32: 128 chunks, threads 0..127 load K, threads 128..255 load V - every thread has exactly one load,
    // so neither case needs a condition around it.
    // Two register sets: tile t+2 is requested at the start of tile t and tile t+1 (requested a tile earlier) is written to LDS
    // at its end - a load has two tiles to arrive.  FAST (the tile is full): no condition around the loads; the bulk of the
    // sweep uses only that form, so hipcc can count the loads in flight and waits for the older tile alone.
    constexpr int NCH = kKT2 * CPR;                      // chunks per operand tile
    constexpr bool HALF = 2 * NCH == THREADS;            // first half of the workgroup loads K, second half V
    static_assert(HALF || NCH == THREADS, "one or two chunks per thread");
    const int ltid = HALF ? (tid % NCH) : tid;
    const bool isv = HALF && tid >= NCH;
    const int lrow = ltid / CPR, lch = ltid % CPR;
    const T* xbase = isv ? vp : kp;                      // HALF: this thread's operand
    const int64_t xsn = isv ? a.v_sn : a.k_sn;
    const T* knext = xbase + (int64_t)lrow * xsn + lch * 8;
    const T* vnext = vp + (int64_t)lrow * a.v_sn + lch * 8;
    const int64_t kstep = (int64_t)kKT2 * xsn, vstep = (int64_t)kKT2 * a.v_sn;
    Chunk8<T> kreg[2], vreg[2];
    auto issue = [&](auto set_tag, auto fast_tag, int row0) {
        constexpr int set = decltype(set_tag)::value;
        if (decltype(fast_tag)::value || row0 + kKT2 <= a.Nk) {
            kreg[set] = load_chunk<T>(knext, 8, true);
            if constexpr (!HALF) vreg[set] = load_chunk<T>(vnext, 8, true);
        } else {
            kreg[set] = load_row_chunk<T, true>(xbase, xsn, row0 + lrow, a.Nk, lch * 8);
            if constexpr (!HALF) vreg[set] = load_row_chunk<T, true>(vp, a.v_sn, row0 + lrow, a.Nk, lch * 8);
        }
        knext += kstep;
        vnext += vstep;
    };
    auto commit = [&](auto set_tag, int buf) {
        constexpr int set = decltype(set_tag)::value;
        bf16x8 im[1];
        chunk_split<T>(kreg[set], im);
        tile_store<D>(lds + (buf * 2 + (isv ? 1 : 0)) * TILE, lrow, lch, im[0]);
        if constexpr (!HALF) {
            chunk_split<T>(vreg[set], im);
            tile_store<D>(lds + (buf * 2 + 1) * TILE, lrow, lch, im[0]);
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    const int nt = (a.Nk + kKT2 - 1) / kKT2;
    issue(B0{}, std::false_type{}, 0);
    commit(B0{}, 0);
    if (nt > 1) issue(B1{}, std::false_type{}, kKT2);
    __syncthreads();

    float l[2] = {0.f, 0.f};
    f32x16 negm[2], o[2][D / 32];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[blk][i] = 0.f;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[blk][dt][i] = 0.f;
    }
    const bf16* kaddr[D / 16];
    const bf16* vaddr[D / 32][2];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) kaddr[s] = lds + row_frag_lane_off<D>(16 * s, lane);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
        int oa, ob;
        tr_frag_lane_off<D, true>(32 * dt, lane, oa, ob);
        vaddr[dt][0] = lds + oa;
        vaddr[dt][1] = lds + ob;
    }

    auto tile = [&](auto mask_tag, auto buf_tag, auto bulk_tag, int t) {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr int buf = decltype(buf_tag)::value;            // = t & 1 = which half of the 64-key hash tile
        constexpr bool BULK = decltype(bulk_tag)::value;         // tile t+2 exists and is full (the caller knows)
        constexpr int KOFF = (buf * 2 + 0) * TILE, VOFF = (buf * 2 + 1) * TILE;
        if (BULK || t + 2 < nt) issue(buf_tag, bulk_tag, (t + 2) * kKT2);
        f32x16 st[2];
        auto scores = [&]() {                                     // S^T = K Q^T in exp2 units, minus the reference exponent
            st[0] = negm[0];
            st[1] = negm[1];
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kaddr[s] + KOFF);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) st[blk] = mfma32(kf, qf[blk][0][s], st[blk]);
            }
            if constexpr (MASK) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (t * kKT2 + acc_row(i, h) >= a.Nk) st[blk][i] = -INFINITY;
            }
        };
        float rs[2];
        auto exps = [&]() {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                // two plain add chains per block: hipcc otherwise pairs the two blocks' sums into one dependent v_pk_add_f32
                // chain (1.5 issue slots per instruction plus wait states between dependent packed ops)
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(st[blk][i]), p1 = __builtin_amdgcn_exp2f(st[blk][i + 1]);
                    acc0 += p0;
                    asm("" : "+v"(acc0));       // opaque to the SLP vectoriser; the add itself stays a compiler instruction (hazard handling)
                    acc1 += p1;
                    asm("" : "+v"(acc1));
                    st[blk][i] = p0;
                    st[blk][i + 1] = p1;
                }
                rs[blk] = acc0 + acc1;
            }
        };
        scores();
        // Steady state: no row maximum at all.  The exponentials are taken against the current reference and their row
        // sums (needed anyway) tell whether some score outgrew it: a probability above 2^kRescaleLog2 pushes its half-row
        // sum past that bound, an overflow makes it inf, a NaN fails the comparison.  Only then (and on the first tile, which
        // has no reference yet) are the scores rebuilt and the reference moved to the new maximum.
        // (Without dropout the loop has issue slots to spare and the classic test on the tile's row maximum is 3 % faster.)
        bool move_ref = t == 0;
        if constexpr (DROP) {
            if (!move_ref) {
                exps();
                move_ref = __any(!(fmaxf(rs[0], rs[1]) <= kRescaleSum));
                if (move_ref) scores();
            }
        } else {
            if (!move_ref) {
                float mloc[2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    float m = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = fmaxf(m, st[blk][i]);
                    mloc[blk] = fmaxf(m, __shfl_xor(m, 32, 64));
                }
                move_ref = __any(fmaxf(mloc[0], mloc[1]) > kRescaleLog2);
            }
        }
        if (move_ref) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, st[blk][i]);
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const float shift = t == 0 ? m : fmaxf(m, 0.f);
                const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-shift);
                l[blk] *= alpha;
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[blk][dt][i] *= alpha;
                const float nm = negm[blk][0] - shift;      // the reference exponent lives (negated) in the seed block only
#pragma unroll
                for (int i = 0; i < 16; ++i) negm[blk][i] = nm;
#pragma unroll
                for (int i = 0; i < 16; ++i) st[blk][i] -= shift;
            }
        }
        if (!DROP || move_ref) exps();
        l[0] += rs[0];
        l[1] += rs[1];
        // O^T[d][q] += V^T P^T: each transposed V fragment serves both query blocks
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 vfr[D / 32];
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) vfr[dt] = tr_frag_at(vaddr[dt][0] + (VOFF + 16 * s2 * D), vaddr[dt][1] + (VOFF + 16 * s2 * D));
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = st[blk][8 * s2 + j];
                bf16x8 pf[1];
                acc_split<1>(x, pf);
                if constexpr (DROP) {
                    const uint32_t rk_tile = rowkey[blk] + (uint32_t)(t >> 1) * kTileAdd;
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {      // group j = 8 buf + 2 (2 s2 + g2) + h of hash tile t >> 1
                        const uint32_t m = rk_tile ^ (drop_grp_a(buf) ^ drop_grp_b(2 * s2 + g2));
                        uint32_t la, lb;
                        drop_lots4(m, la, lb);
                        const uint32_t ma = drop_keepmask2(la, tm1x2), mb = drop_keepmask2(lb, tm1x2);
                        u32x4 w = __builtin_bit_cast(u32x4, pf[0]);
                        w[2 * g2] &= ma;
                        w[2 * g2 + 1] &= mb;
                        pf[0] = __builtin_bit_cast(bf16x8, w);
                    }
                }
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) o[blk][dt] = mfma32(vfr[dt], pf[0], o[blk][dt]);
            }
        }
        if (BULK || t + 1 < nt) commit(std::integral_constant<int, buf ^ 1>{}, buf ^ 1);
        __syncthreads();
    };
    const bool ragged = (a.Nk % kKT2) != 0;
    const int nfull = ragged ? nt - 1 : nt;
    int t = 0;
    for (; t + 3 < nfull; t += 2) {                             // tiles t+2 and t+3 are full: branch-free requests
        tile(std::false_type{}, B0{}, std::true_type{}, t);
        tile(std::false_type{}, B1{}, std::true_type{}, t + 1);
    }
    for (; t + 1 < nfull; t += 2) {
        tile(std::false_type{}, B0{}, std::false_type{}, t);
        tile(std::false_type{}, B1{}, std::false_type{}, t + 1);
    }
    if (t < nfull) tile(std::false_type{}, B0{}, std::false_type{}, t);            // t is even here
    if (ragged) {
        if ((nt - 1) & 1) tile(std::true_type{}, B1{}, std::false_type{}, nt - 1);
        else tile(std::true_type{}, B0{}, std::false_type{}, nt - 1);
    }

#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const float ltot = l[blk] + __shfl_xor(l[blk], 32, 64);
        const float inv = (DROP ? a.keep_scale : 1.f) / ltot;
        const int qrow = qb * QBW + (int)(threadIdx.x >> 6) * 64 + 32 * blk + (int)(threadIdx.x & 31);      // recomputed: not kept live through the sweep
        if (qrow < a.Nq) {
            T* orow = op + (int64_t)qrow * a.o_sn;
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = f2bf(o[blk][dt][4 * g + j] * inv);
                    *reinterpret_cast<bf16x4*>(orow + 32 * dt + 8 * g + 4 * h) = w;
                }
            if (h == 0) a.lse[(int64_t)bh * a.Nq + qrow] = (__builtin_amdgcn_logf(ltot) - negm[blk][0]) * kLn2;
        }
    }
}

// ---------------------------------------------------------------------------------
// forward, software-pipelined (round 4; bf16, 16-byte addressable operands, LDS-DMA tiles)
// ---------------------------------------------------------------------------------
// Same arithmetic as attn_fwd_kernel (32 query rows per wavefront, 64-key tiles, S^T = K Q^T with the query on the lane, the same
// dropout lots), but the tile loop is skewed by one tile so that every stretch of the steady state carries matrix AND vector
// work that do not depend on each other - an in-order wavefront can then issue its exponentials / hashes / converts in the
// shadow of its own MFMAs instead of running them phase after phase (the phase-separated kernels above overlap matrix and vector
// work only between the two wavefronts of a SIMD, which one barrier per tile keeps in the SAME phase; rocprofv3 of round 3: matrix
// pipe 41 % busy, vector issue 48 %, 63 % of the wave-cycles waiting):
//   block 1 of iteration t :  S(t+1) = K(t+1) Q^T - m   (8 MFMA + 8 K row-fragment reads)   ||  p = exp2(S(t)), row sums (64 VALU)
//   block 2 of iteration t :  O += V(t)^T P(t)^T        (8 MFMA + 16 transposed V reads)    ||  convert + keep-mask hash of the next
//                             16-key group of P(t) (the hash is independent of any data)
// There is NO branch in the steady state.  The reference exponent m of a row is fixed by tile 0 (every row adopts that tile's
// maximum) and is never moved afterwards: probabilities are kept as exp2(s - m) in fp32 / bf16, whose exponent range (2^127) -
// not the 2^6 head-room of the kernels above - is the only limit, and the relative precision of P, of the row sum and of O does not
// depend on the common factor 2^(max - m) (it cancels in O = sum P V / sum P; LSE = m + log2 sum P).  A row whose later scores
// exceed tile 0's maximum by more than ~2^64 (44 nats: never seen outside adversarial tests) shows in its row sum (> 2^80, inf or
// NaN); its wavefront then recomputes its 32 rows after the sweep with the classic online softmax straight from global
// memory (slow_rows below: same MFMA fragments, no LDS, no barriers - correct, 3-5x slower, exercised by the spike tests).
// (A rescale path that rejoins the loop cost ~95 register copies per tile on the COMMON path - hipcc places the copies of the
// join there - or, under __builtin_expect, had the block outlined as a cold function with the whole loop state in scratch.)
// K and V are kept four tiles deep in LDS each (read / landed / in flight / free), fetched by LDS-DMA TWO tiles ahead of their
// readers from a wave-uniform base + 32-bit offsets behind a COUNTED vmcnt (the newest requests stay in flight across the one
// barrier per tile: a request has two iterations, ~1 us, to land - with one iteration every tile ended waiting for it).
constexpr int kKBufs = 4;
constexpr float kOverflowSum = 1.2089258e24f;      // 2^80: a row's probabilities sum to less (so |O| <= 2^80 max|v| stays finite), or the row is redone carefully
template <int D, bool DROP, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void attn_fwdp_kernel(const AttnArgs a_in) {
    using T = bf16;
    constexpr int THREADS = 64 * WAVES, QBW = 32 * WAVES;
    constexpr int TILE = kKT * D;                      // elements per tile image
    constexpr int CPR = D / 8, CH = kKT * CPR;         // 16-byte chunks per tile row / per tile
    constexpr bool SPLIT = 2 * CH == THREADS;          // half of the workgroup fetches K, the other half V (one chunk per thread)
    constexpr int NCH = SPLIT ? 1 : CH / THREADS;      // chunks per thread and operand otherwise
    static_assert(SPLIT || (CH % THREADS == 0 && NCH >= 1), "tile chunks must divide over the workgroup");
    AttnArgs a = a_in;
    a.seed_lo = seed_with_counter(a_in.seed_lo, a_in.seed_ctr);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);         // [4 K tiles][4 V tiles]
    constexpr int VBASE = kKBufs * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int nqb = (a.Nq + QBW - 1) / QBW;
    int bh, qb;
    block_map(blockIdx.x, a.B * a.H, nqb, bh, qb);
    const int b = bh / a.H, hh = bh % a.H;
    const T* qp = reinterpret_cast<const T*>(a.q) + b * a.q_sb + hh * a.q_sh;
    const T* kp = reinterpret_cast<const T*>(a.k) + b * a.k_sb + hh * a.k_sh;
    const T* vp = reinterpret_cast<const T*>(a.v) + b * a.v_sb + hh * a.v_sh;
    T* op = reinterpret_cast<T*>(a.o) + b * a.o_sb + hh * a.o_sh;

    const int qrow = qb * QBW + wave * 32 + r;
    const bool qvalid = qrow < a.Nq;
    const int qrow_c = qvalid ? qrow : a.Nq - 1;
    const float sl2 = a.scale * kLog2e;
    bf16x8 qf[1][D / 16];
    load_row_frags_scaled<T, 1, D / 16, true>(qp + (int64_t)qrow_c * a.q_sn, h, true, sl2, qf);
    const uint32_t rowkey = DROP ? drop_rowkey(a, bh, qrow_c) ^ (h ? kGrpH : 0u) : 0u;
    const uint32_t tm1x2 = DROP ? ((uint32_t)(drop_ts(a) - 1) & 0xffffu) * 0x10001u : 0u;

    // tile loader: thread -> chunk ltid (+ THREADS i) of its operand's tile; the image is lane-linear, the XOR swizzle is applied
    // to the SOURCE chunk (see TileLoader).  Addresses are a wave-uniform base (K / V of this (b, h)) plus a 32-bit byte offset
    // (the launcher checks that a (b, h) slice spans < 2^31 bytes); rows past the end of K / V are clamped to the last row
    // (finite values, zero weight) by clamping the row's byte offset.
    const int ltid = SPLIT ? tid % CH : tid;
    const bool isv = SPLIT && tid >= CH;
    const int lrow = ltid / CPR;
    const uint32_t lch8 = (uint32_t)(tile_off<D>(lrow, ltid % CPR) - lrow * D) * 2u;     // byte offset of the logical chunk in its row
    const int dwave = __builtin_amdgcn_readfirstlane((ltid >> 6) * 64 * 8);
    const uint32_t k_rowb = (uint32_t)a.k_sn * 2u, v_rowb = (uint32_t)a.v_sn * 2u;      // bytes per row
    const uint32_t k_last = (uint32_t)(a.Nk - 1) * k_rowb, v_last = (uint32_t)(a.Nk - 1) * v_rowb;
    uint32_t k_roff[NCH], v_roff[NCH];                                                   // byte offsets of this thread's rows in tile 0
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        k_roff[i] = (uint32_t)(lrow + i * (THREADS / CPR)) * k_rowb;
        v_roff[i] = (uint32_t)(lrow + i * (THREADS / CPR)) * v_rowb;
    }
    auto dma_k = [&](int tile, bf16* image) {
        const uint32_t toff = (uint32_t)tile * (uint32_t)kKT * k_rowb;                  // wave-uniform
#pragma unroll
        for (int i = 0; i < NCH; ++i) lds_dma16_s(kp, min(k_roff[i] + toff, k_last) + lch8, image + (i * THREADS * 8 + dwave));
    };
    auto dma_v = [&](int tile, bf16* image) {
        const uint32_t toff = (uint32_t)tile * (uint32_t)kKT * v_rowb;
#pragma unroll
        for (int i = 0; i < NCH; ++i) lds_dma16_s(vp, min(v_roff[i] + toff, v_last) + lch8, image + (i * THREADS * 8 + dwave));
    };
    auto fetch = [&](int kt_tile, bf16* kimage, int vt_tile, bf16* vimage) {      // K tile kt_tile and V tile vt_tile
        if constexpr (SPLIT) {
            if (isv) dma_v(vt_tile, vimage);
            else dma_k(kt_tile, kimage);
        } else {
            dma_k(kt_tile, kimage);
            dma_v(vt_tile, vimage);
        }
    };
    // every DMA but the pieces this thread issued in the CURRENT iteration has landed (vmcnt counts in issue order): the tiles the
    // next iteration reads were requested a whole iteration earlier, the newest requests stay in flight across the barrier
    constexpr int INFLIGHT = SPLIT ? 1 : 2 * NCH;
    auto dma_wait_prev = [] { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory"); };
    auto dma_wait_all = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    const bf16* kaddr[D / 16];
    const bf16* vaddr[D / 32][2];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) kaddr[s] = lds + row_frag_lane_off<D>(16 * s, lane);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
        int oa, ob;
        tr_frag_lane_off<D, true>(32 * dt, lane, oa, ob);
        vaddr[dt][0] = lds + oa;
        vaddr[dt][1] = lds + ob;
    }

    const int nt = (a.Nk + kKT - 1) / kKT;
    // prologue: K(0), K(1), K(2), V(0), V(1)
    if constexpr (SPLIT) {
        if (isv) { dma_v(0, lds + VBASE); dma_v(1, lds + VBASE + TILE); }
        else { dma_k(0, lds); dma_k(1, lds + TILE); dma_k(2, lds + 2 * TILE); }
    } else {
        dma_k(0, lds);
        dma_k(1, lds + TILE);
        dma_k(2, lds + 2 * TILE);
        dma_v(0, lds + VBASE);
        dma_v(1, lds + VBASE + TILE);
    }
    dma_wait_all();
    __syncthreads();

    f32x16 negm, o[D / 32];
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float l = 0.f;
    // (Row sums through a ones-operand MFMA - 4 more MFMAs per tile instead of 32 vector adds - measured 2 % SLOWER: matrix time is
    // not free under the vector stream here; profiles/r04_attention_pipelined_forward.txt.)

    auto mask_tail = [&](f32x16 (&st)[2], int t) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (t * kKT + 32 * kt + acc_row(i, h) >= a.Nk) st[kt][i] = -INFINITY;
    };
    auto row_max = [&](const f32x16 (&st)[2]) {
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, st[kt][i]);
        return fmaxf(m, __shfl_xor(m, 32, 64));
    };
    // p = exp2(s) for elements [e0, e0 + n) of the tile (element = 16 kt + i), in place, summed into two plain add chains that are
    // opaque to the SLP vectoriser (see attn_fwd2_kernel); a tile's chains are folded into l by close_sums()
    // (sums as four packed v_pk_add_f32 chains - half the add instructions - measured 1 - 3 % slower: a packed fp32 op holds the vector
    // pipe twice as long)
    float acc0 = 0.f, acc1 = 0.f;
    auto exps = [&](f32x16 (&st)[2], int e0, int n) {
#pragma unroll
        for (int e = e0; e < e0 + n; e += 2) {
            const int kt = e >> 4, i = e & 15;
            const float p0 = __builtin_amdgcn_exp2f(st[kt][i]), p1 = __builtin_amdgcn_exp2f(st[kt][i + 1]);
            acc0 += p0;
            asm("" : "+v"(acc0));
            acc1 += p1;
            asm("" : "+v"(acc1));
            st[kt][i] = p0;
            st[kt][i + 1] = p1;
        }
    };
    auto close_sums = [&] {
        l += acc0 + acc1;
        acc0 = 0.f;
        acc1 = 0.f;
    };
    // O^T += V^T P^T for the probabilities in st of tile `tile`, V image at element offset voff: per 16-key group g = 2 kt + s2 the
    // probabilities are converted to bf16 and masked (keep masks hashed right here: independent of any data) and the transposed V
    // fragments requested while the MFMAs of the group before run; the fences keep hipcc from hoisting every LDS read to the top
    // (it then spilt the Q fragments into scratch inside the loop) and from clustering the MFMAs at the end.
    auto pv = [&](const f32x16 (&st)[2], int tile, int voff) {
        const uint32_t rk_tile = rowkey + (uint32_t)tile * kTileAdd;
        auto prep = [&](int g, bf16x8& pf, bf16x8 (&vf)[D / 32]) {
            const int kt = g >> 1, s2 = g & 1;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = st[kt][8 * s2 + j];
            bf16x8 im[1];
            acc_split<1>(x, im);
            if constexpr (DROP) {
                u32x4 w = __builtin_bit_cast(u32x4, im[0]);
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {      // group j = 8 kt + 2 (2 s2 + g2) + h of the tile
                    const uint32_t m = rk_tile ^ (drop_grp_a(kt) ^ drop_grp_b(2 * s2 + g2));
                    uint32_t la, lb;
                    drop_lots4(m, la, lb);
                    w[2 * g2] &= drop_keepmask2(la, tm1x2);
                    w[2 * g2 + 1] &= drop_keepmask2(lb, tm1x2);
                }
                im[0] = __builtin_bit_cast(bf16x8, w);
            }
            pf = im[0];
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) {
                const int VO = voff + (32 * kt + 16 * s2) * D;
                vf[dt] = tr_frag_at(vaddr[dt][0] + VO, vaddr[dt][1] + VO);
            }
        };
        bf16x8 pf, pn, vf[D / 32], vn[D / 32];
        prep(0, pf, vf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) o[dt] = mfma32(vf[dt], pf, o[dt]);
            if (g + 1 < 4) prep(g + 1, pn, vn);
            pf = pn;
#pragma unroll
            for (int dt = 0; dt < D / 32; ++dt) vf[dt] = vn[dt];
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // tile 0: its scores and the reference (every row adopts this tile's maximum, for the whole sweep)
    f32x16 sc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        sc[kt] = negm;
#pragma unroll
        for (int s = 0; s < D / 16; ++s)
            sc[kt] = mfma32(*reinterpret_cast<const bf16x8*>(kaddr[s] + 32 * kt * D), qf[0][s], sc[kt]);
    }
    if (a.Nk < kKT) mask_tail(sc, 0);
    {
        const float m0 = row_max(sc);
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = -m0;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[kt][i] -= m0;
    }

    // steady state, tile t (T4 = t & 3 selects the buffers at compile time); tile t+1 exists
    auto body = [&](auto t4_tag, int t) {
        constexpr int T4 = decltype(t4_tag)::value;
        constexpr int KR = ((T4 + 1) & 3) * TILE, KD = ((T4 + 3) & 3) * TILE;
        constexpr int VR = VBASE + T4 * TILE, VD = VBASE + ((T4 + 2) & 3) * TILE;
        fetch(t + 3, lds + KD, t + 2, lds + VD);         // two tiles ahead of their readers
        // block 1: S(t+1) = K(t+1) Q^T - m, one 16-wide d step (2 MFMAs) at a time, each with its share of tile t's exponentials
        // behind it and the next step's two K fragments requested ahead
        f32x16 sn[2];
        bf16x8 kf[2], kn[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[kt] = *reinterpret_cast<const bf16x8*>(kaddr[0] + (KR + 32 * kt * D));
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            if (s + 1 < D / 16) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) kn[kt] = *reinterpret_cast<const bf16x8*>(kaddr[s + 1] + (KR + 32 * kt * D));
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) sn[kt] = mfma32(kf[kt], qf[0][s], s == 0 ? negm : sn[kt]);
            constexpr int EPS = 32 / (D / 16);            // exponentials per d step (8 at d = 64, 16 at d = 32)
            exps(sc, s * EPS, EPS);
            kf[0] = kn[0];
            kf[1] = kn[1];
            __builtin_amdgcn_sched_barrier(0);
        }
        close_sums();
        pv(sc, t, VR);                                    // block 2
        sc[0] = sn[0];
        sc[1] = sn[1];
        dma_wait_prev();
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    int t = 0;
    for (; t + 4 < nt; t += 4) {
        body(I0{}, t);
        body(I1{}, t + 1);
        body(I2{}, t + 2);
        body(I3{}, t + 3);
    }
    if (t + 1 < nt) {
        body(I0{}, t);
        ++t;
        if (t + 1 < nt) {
            body(I1{}, t);
            ++t;
            if (t + 1 < nt) {
                body(I2{}, t);
                ++t;
            }
        }
    }
    // last tile (t = nt - 1): nothing to prefetch, ragged tail masked
    if (a.Nk % kKT) mask_tail(sc, t);
    exps(sc, 0, 32);
    close_sums();
    pv(sc, t, VBASE + (t & 3) * TILE);
    dma_wait_all();                                       // the (clamped, unused) requests of the last iterations: nothing in flight at exit

    if (__any(!(l <= kOverflowSum))) {
        // Some row of this wavefront outgrew tile 0's reference by more than the fp32 exponent range allows (or produced a NaN):
        // its 32 rows again, carefully - classic online softmax with the reference moved BEFORE the exponentials (as in
        // attn_fwd_kernel), operands straight from global memory in fragment layout (K rows: one 16-byte load per lane; V^T:
        // eight 2-byte loads per fragment), no LDS (other wavefronts may still be reading the last tiles), no barriers.
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = 0.f;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
        l = 0.f;
        for (int ts = 0; ts < nt; ++ts) {
            f32x16 st[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                st[kt] = negm;
                const int key = min(ts * kKT + 32 * kt + r, a.Nk - 1);
#pragma unroll
                for (int s = 0; s < D / 16; ++s) {
                    const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(kp + (int64_t)key * a.k_sn + 16 * s + 8 * h);
                    st[kt] = mfma32(kfr, qf[0][s], st[kt]);
                }
            }
            if ((ts + 1) * kKT > a.Nk) mask_tail(st, ts);
            const float mloc = row_max(st);
            if (ts == 0 || __any(mloc > kRescaleLog2)) {
                const float shift = ts == 0 ? mloc : fmaxf(mloc, 0.f);
                const float alpha = ts == 0 ? 1.f : __builtin_amdgcn_exp2f(-shift);
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt)
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
                    st[kt][i] = __builtin_amdgcn_exp2f(st[kt][i]);
                    rs += st[kt][i];
                }
            l += rs;
            const uint32_t rk_tile = rowkey + (uint32_t)ts * kTileAdd;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = st[kt][8 * s2 + j];
                    bf16x8 im[1];
                    acc_split<1>(x, im);
                    if constexpr (DROP) {
                        u32x4 w = __builtin_bit_cast(u32x4, im[0]);
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) {
                            const uint32_t m = rk_tile ^ (drop_grp_a(kt) ^ drop_grp_b(2 * s2 + g2));
                            uint32_t la, lb;
                            drop_lots4(m, la, lb);
                            w[2 * g2] &= drop_keepmask2(la, tm1x2);
                            w[2 * g2 + 1] &= drop_keepmask2(lb, tm1x2);
                        }
                        im[0] = __builtin_bit_cast(bf16x8, w);
                    }
                    const int k0 = ts * kKT + 32 * kt + 16 * s2;      // V^T fragment: rows k0 + 4h + j and k0 + 8 + 4h + j, column 32 dt + r
#pragma unroll
                    for (int dt = 0; dt < D / 32; ++dt) {
                        bf16x8 vfr;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int key = min(k0 + 4 * h + (j & 3) + 8 * (j >> 2), a.Nk - 1);
                            vfr[j] = vp[(int64_t)key * a.v_sn + 32 * dt + r];
                        }
                        o[dt] = mfma32(vfr, im[0], o[dt]);
                    }
                }
        }
    }

    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = (DROP ? a.keep_scale : 1.f) / ltot;
    if (qvalid) {
        T* orow = op + (int64_t)qrow * a.o_sn;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
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

// ---------------------------------------------------------------------------------
// backward: delta[bh][q] = sum_d dO * O
// ---------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs a) {
    constexpr int LPR = D / 8;                      // lanes per row
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = gid / LPR;
    const int part = (int)(gid % LPR);
    const int64_t nrows = (int64_t)a.B * a.H * a.Nq;
    float acc = 0.f;
    if (row < nrows) {
        const int q = (int)(row % a.Nq);
        const int bh = (int)(row / a.Nq);
        const int b = bh / a.H, hh = bh % a.H;
        const T* op = reinterpret_cast<const T*>(a.o) + b * a.o_sb + hh * a.o_sh + (int64_t)q * a.o_sn + part * 8;
        const T* dp = reinterpret_cast<const T*>(a.dout) + b * a.do_sb + hh * a.do_sh + (int64_t)q * a.do_sn + part * 8;
        Chunk8<T> x = load_chunk<T>(op, 8, a.vec != 0);
        Chunk8<T> y = load_chunk<T>(dp, 8, a.vec != 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += chunk_get<T>(x, j) * chunk_get<T>(y, j);
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (row < nrows && part == 0) a.delta[row] = acc;
}

// ---------------------------------------------------------------------------------
// backward: dQ
// ---------------------------------------------------------------------------------
// WAVES = 8 (round 3; chosen by launch_bwd): 256 query rows per workgroup, one workgroup per CU; the K / V tiles are staged once for eight
// wavefronts (by wavefronts 0-3) - see attn_bwd_dkv_kernel.
template <typename T, int D, bool DROP, bool VEC, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? (D == 32 && sizeof(T) == 2 ? 3 : 2) : 1) void attn_bwd_dq_kernel(const AttnArgs a_in) {
    constexpr int QB = 32 * WAVES;
    AttnArgs a = a_in;
    a.seed_lo = seed_with_counter(a_in.seed_lo, a_in.seed_ctr);
    constexpr int NS = NSplit<T>::value;
    constexpr int TILE = kKT * D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bool loader = WAVES == 4 || tid < 256;

    const int nqb = (a.Nq + QB - 1) / QB;
    int bh, qb;
    block_map(blockIdx.x, a.B * a.H, nqb, bh, qb);
    const int b = bh / a.H, hh = bh % a.H;
    const T* qp = reinterpret_cast<const T*>(a.q) + b * a.q_sb + hh * a.q_sh;
    const T* kp = reinterpret_cast<const T*>(a.k) + b * a.k_sb + hh * a.k_sh;
    const T* vp = reinterpret_cast<const T*>(a.v) + b * a.v_sb + hh * a.v_sh;
    const T* dop = reinterpret_cast<const T*>(a.dout) + b * a.do_sb + hh * a.do_sh;
    T* dqp = reinterpret_cast<T*>(a.dq) + b * a.dq_sb + hh * a.dq_sh;

    const int q0 = qb * QB + wave * 32;
    const int qrow = q0 + r;
    const bool qvalid = qrow < a.Nq;
    const int qrow_c = qvalid ? qrow : a.Nq - 1;

    // Q rows carry scale * log2(e); the S^T / dP^T accumulators start from -lse2[q] / -delta[q] (row constants of
    // this lane's query), so the chains deliver s - lse2 and dP - delta without per-element arithmetic.
    const float sl2 = a.scale * kLog2e;
    bf16x8 qf[NS][D / 16], dof[NS][D / 16];
    load_row_frags_scaled<T, NS, D / 16, VEC>(qp + (int64_t)qrow_c * a.q_sn, h, true, sl2, qf);
    load_row_frags<T, NS, D / 16, VEC>(dop + (int64_t)qrow_c * a.do_sn, h, true, dof);
    const float nlse2 = -a.lse[(int64_t)bh * a.Nq + qrow_c] * kLog2e;
    const float ndelta = -a.delta[(int64_t)bh * a.Nq + qrow_c] * (DROP ? 1.f / a.keep_scale : 1.f);
    f32x16 c_s, c_dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c_s[i] = nlse2; c_dp[i] = ndelta; }

    TileLoader<T, D, VEC> kl, vl;
    auto Kt = [&](int buf) { return lds + (buf * 2 + 0) * NS * TILE; };
    auto Vt = [&](int buf) { return lds + (buf * 2 + 1) * NS * TILE; };
    const int nt = (a.Nk + kKT - 1) / kKT;
    if (loader) {
        kl.init(kp, a.k_sn, 0, tid);
        vl.init(vp, a.v_sn, 0, tid);
        kl.issue(kp, a.k_sn, 0, a.Nk, tid, Kt(0));
        vl.issue(vp, a.v_sn, 0, a.Nk, tid, Vt(0));
        kl.commit(Kt(0), tid);
        vl.commit(Vt(0), tid);
    }
    kl.wait();
    __syncthreads();

    f32x16 dq[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[dt][i] = 0.f;
    const uint32_t rowkey = DROP ? drop_rowkey(a, bh, qrow_c) ^ (h ? kGrpH : 0u) : 0u;      // lane half = low bit of the group index
    const int ts = drop_ts(a);

    // per-lane LDS addresses (see the forward kernel): K / V row fragments share one set, K^T fragments another
    const bf16* raddr[D / 16];
    const bf16* taddr[D / 32][2];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) raddr[s] = lds + row_frag_lane_off<D>(16 * s, lane);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
        int oa, ob;
        tr_frag_lane_off<D, true>(32 * dt, lane, oa, ob);
        taddr[dt][0] = lds + oa;
        taddr[dt][1] = lds + ob;
    }

    auto tile = [&](auto mask_tag, auto buf_tag, int t) {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr int buf = decltype(buf_tag)::value;
        constexpr int KOFF = (buf * 2 + 0) * NS * TILE, VOFF = (buf * 2 + 1) * NS * TILE;
        if (t + 1 < nt && loader) {
            kl.issue(kp, a.k_sn, (t + 1) * kKT, a.Nk, tid, Kt(buf ^ 1));
            vl.issue(vp, a.v_sn, (t + 1) * kKT, a.Nk, tid, Vt(buf ^ 1));
        }
        f32x16 st[2], dpt[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[kt] = c_s;
            dpt[kt] = c_dp;
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
#pragma unroll
                for (int sa = 0; sa < NS; ++sa) {
                    bf16x8 kf = *reinterpret_cast<const bf16x8*>(raddr[s] + (KOFF + sa * TILE + 32 * kt * D));
                    bf16x8 vf = *reinterpret_cast<const bf16x8*>(raddr[s] + (VOFF + sa * TILE + 32 * kt * D));
#pragma unroll
                    for (int sb = 0; sb < NS; ++sb)
                        if (sa + sb <= 1) {
                            st[kt] = mfma32(kf, qf[sb][s], st[kt]);
                            dpt[kt] = mfma32(vf, dof[sb][s], dpt[kt]);
                        }
                }
            }
        }
        const int kbase = t * kKT;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kt][i] = __builtin_amdgcn_exp2f(st[kt][i]);
        // dS / keep_scale = p * (keep * dP - delta / keep_scale); keep_scale rejoins in the epilogue
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const int key = kbase + 32 * kt + acc_row(i, h);
                float dsel[4] = {dpt[kt][i], dpt[kt][i + 1], dpt[kt][i + 2], dpt[kt][i + 3]};                       // keep * dP - delta
                if constexpr (DROP) drop_select4((rowkey + (uint32_t)t * kTileAdd) ^ (drop_grp_a(kt) ^ drop_grp_b(i >> 2)), ts, dsel, ndelta, dsel);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    f32x2 p2 = {st[kt][i + j], st[kt][i + j + 1]};
                    if constexpr (MASK) {
                        if (key + j >= a.Nk) p2[0] = 0.f;
                        if (key + j + 1 >= a.Nk) p2[1] = 0.f;
                    }
                    const f32x2 d2 = {dsel[j], dsel[j + 1]};
                    const f32x2 r2 = p2 * d2;
                    st[kt][i + j] = r2[0];
                    st[kt][i + j + 1] = r2[1];
                }
            }
        // dQ[q][d] += dS K  (dS^T accumulators as the A operand, K through the transposed read)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = st[kt][8 * s2 + j];
                bf16x8 dsf[NS];
                acc_split<NS>(x, dsf);
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
                    for (int sb = 0; sb < NS; ++sb) {
                        const int KO = KOFF + sb * TILE + (32 * kt + 16 * s2) * D;
                        bf16x8 kf = tr_frag_at(taddr[dt][0] + KO, taddr[dt][1] + KO);
#pragma unroll
                        for (int sa = 0; sa < NS; ++sa)
                            if (sa + sb <= 1) dq[dt] = mfma32(dsf[sa], kf, dq[dt]);
                    }
                }
            }
        }
        if (t + 1 < nt && loader) {
            kl.commit(Kt(buf ^ 1), tid);
            vl.commit(Vt(buf ^ 1), tid);
        }
        kl.wait();
        __syncthreads();
    };
    const bool ragged = (a.Nk % kKT) != 0;
    const int nfull = ragged ? nt - 1 : nt;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 1 < nfull; t += 2) {
        tile(std::false_type{}, B0{}, t);
        tile(std::false_type{}, B1{}, t + 1);
    }
    if (t < nfull) tile(std::false_type{}, B0{}, t);            // t is even here
    if (ragged) {
        if ((nt - 1) & 1) tile(std::true_type{}, B1{}, nt - 1);
        else tile(std::true_type{}, B0{}, nt - 1);
    }
    const float oscale = a.scale * (DROP ? a.keep_scale : 1.f);
    // dq tile: rows = q (registers), col = d (lane)
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qq = q0 + acc_row(i, h);
            if (qq < a.Nq) dqp[(int64_t)qq * a.dq_sn + 32 * dt + r] = from_f<T>(dq[dt][i] * oscale);
        }
}

// ---------------------------------------------------------------------------------
// backward: dK, dV
// ---------------------------------------------------------------------------------
// (Round 4 built a software-pipelined form of this kernel - the dV / dK products of slice j-1 issued under the vector work of slice
// j, Q / dO four tiles deep by LDS-DMA, commits afce664 and (with every LDS operand requested a sub-step ahead) its successor -
// bit-compatible lots, parity green, and +-0 % on every BASELINE shape
// (profiles/r04_attention_pipelined_dkv_experiment.txt): with 438 instructions per 32 MFMAs the kernel is bound by the rate at
// which a SIMD issues instructions of ANY kind (~5 cycles each with two wavefronts), not by the order they are issued in.)
// WAVES = 4: 128 keys per workgroup, two workgroups per CU.  WAVES = 8 (round 3; chosen by launch_bwd): 256 keys per
// workgroup, one per CU - the Q / dO tiles, row constants and lots are staged once for eight wavefronts instead of four (by
// wavefronts 0-3, whose loader instructions also set them half a step behind 4-7), one barrier keeps the two wavefronts of a SIMD
// in a fixed phase.
template <typename T, int D, bool DROP, bool VEC, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? (D == 32 && sizeof(T) == 2 ? 3 : 2) : 1) void attn_bwd_dkv_kernel(const AttnArgs a_in) {
    constexpr int KB = 32 * WAVES;                  // keys per workgroup
    constexpr int LP = LotTile<WAVES>::PART, LS = LotTile<WAVES>::ROW;      // lots per part / per tile row
    static_assert(WAVES == 4 || WAVES == 8, "4 or 8 wavefronts");
    AttnArgs a = a_in;
    a.seed_lo = seed_with_counter(a_in.seed_lo, a_in.seed_ctr);
    constexpr int NS = NSplit<T>::value;
    constexpr int TILE = kKT * D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);                       // [buf][Q|dO][NS][TILE]
    float* stat = reinterpret_cast<float*>(lds + 4 * NS * TILE);     // [buf][lse2|delta][kKT]
    // [buf][64 q][LS] 16-bit dropout lots of the current q-tile x this workgroup's keys, generated cooperatively
    // (one hash per 4 keys, as in the forward) so that the per-element cost is a 2-byte LDS read + compare + selects.
    uint16_t* lots = reinterpret_cast<uint16_t*>(stat + 2 * 2 * kKT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = WAVES == 4 || tid < 256;    // the tile loaders are written for 256 threads
    const int r = lane & 31, h = lane >> 5;

    const int nkb = (a.Nk + KB - 1) / KB;
    const int nwg = nkb * a.B * a.H;
    const int split = blockIdx.x / nwg;          // query-range slice (0 when qsplit == 1)
    int bh, kb;
    block_map(blockIdx.x % nwg, a.B * a.H, nkb, bh, kb);
    const int b = bh / a.H, hh = bh % a.H;
    const T* qp = reinterpret_cast<const T*>(a.q) + b * a.q_sb + hh * a.q_sh;
    const T* kp = reinterpret_cast<const T*>(a.k) + b * a.k_sb + hh * a.k_sh;
    const T* vp = reinterpret_cast<const T*>(a.v) + b * a.v_sb + hh * a.v_sh;
    const T* dop = reinterpret_cast<const T*>(a.dout) + b * a.do_sb + hh * a.do_sh;
    T* dkp = reinterpret_cast<T*>(a.dk) + b * a.dk_sb + hh * a.dk_sh;
    T* dvp = reinterpret_cast<T*>(a.dv) + b * a.dv_sb + hh * a.dv_sh;

    const int k0 = kb * KB + wave * 32;
    const int krow = k0 + r;
    const bool kvalid = krow < a.Nk;
    const int krow_c = kvalid ? krow : a.Nk - 1;

    // K rows carry scale * log2(e) (scores in exp2 units); the S / dP accumulators of a query slice start from
    // -lse2[q] / -delta[q] read from the tile's row constants, so p = exp2(acc) and dP - delta need no arithmetic.
    const float sl2 = a.scale * kLog2e;
    bf16x8 kf[NS][D / 16], vf[NS][D / 16];
    load_row_frags_scaled<T, NS, D / 16, VEC>(kp + (int64_t)krow_c * a.k_sn, h, kvalid, sl2, kf);
    load_row_frags<T, NS, D / 16, VEC>(vp + (int64_t)krow_c * a.v_sn, h, kvalid, vf);

    TileLoader<T, D, VEC> ql, dl;
    auto Qt = [&](int buf) { return lds + (buf * 2 + 0) * NS * TILE; };
    auto Dt = [&](int buf) { return lds + (buf * 2 + 1) * NS * TILE; };
    const int nt_all = (a.Nq + kKT - 1) / kKT;
    const int per_split = (nt_all + a.qsplit - 1) / a.qsplit;
    const int t_begin = split * per_split;
    const int nt = min(nt_all, t_begin + per_split);      // this workgroup sweeps query tiles [t_begin, nt)
    float st_l = 0.f, st_d = 0.f;
    // The row constants are only LOADED at the start of a tile; the arithmetic on them waits until commit time at its end.  (Scaling
    // them right away put an s_waitcnt vmcnt(0) behind the loads - in wavefront 0 only - i.e. the full latency of every load just
    // requested, tile loads included, at the head of each tile, and the other three wavefronts then waited for it at the barrier.)
    bool st_ok = false;
    auto issue_stat = [&](int t) {
        if (tid < kKT) {
            const int q = t * kKT + tid;
            st_ok = q < a.Nq;
            const int qc = st_ok ? q : a.Nq - 1;
            st_l = a.lse[(int64_t)bh * a.Nq + qc];
            st_d = a.delta[(int64_t)bh * a.Nq + qc];
        }
    };
    auto commit_stat = [&](int buf) {
        if (tid < kKT) {
            stat[(buf * 2 + 0) * kKT + tid] = st_ok ? -st_l * kLog2e : -INFINITY;      // out-of-range query rows: p = exp2(-inf) = 0
            stat[(buf * 2 + 1) * kKT + tid] = st_ok ? -st_d * (DROP ? 1.f / a.keep_scale : 1.f) : 0.f;
        }
    };
    // thread -> (q row of the tile, 32-key part of the workgroup's keys): 8 hashes, 32 lots
    auto gen_lots = [&](int t, int buf) {
        if constexpr (DROP) {
            const int ql = tid / WAVES, part = tid % WAVES;
            const int q = min(t * kKT + ql, a.Nq - 1);
            // keys KB kb + 32 part + 4 u .. + 3: 64-key tile (KB / 64) kb + (part >> 1), group j = 8 (part & 1) + u
            const uint32_t rk0 = drop_rowkey(a, bh, q), tadd = (uint32_t)((KB / 64) * kb + (part >> 1)) * kTileAdd;
            const uint32_t rk[2] = {(rk0 + tadd) ^ drop_grp_a(part & 1), ((rk0 ^ kGrpH) + tadd) ^ drop_grp_a(part & 1)};
            uint32_t* dst = reinterpret_cast<uint32_t*>(lots + ((size_t)buf * kKT + ql) * LS + LP * part);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t m = rk[u & 1] ^ drop_grp_b(u >> 1);
                drop_lots4(m, dst[2 * u], dst[2 * u + 1]);
            }
        }
    };
    if (loader) {
        ql.init(qp, a.q_sn, t_begin * kKT, tid);
        dl.init(dop, a.do_sn, t_begin * kKT, tid);
        ql.issue(qp, a.q_sn, t_begin * kKT, a.Nq, tid, Qt(t_begin & 1));
        dl.issue(dop, a.do_sn, t_begin * kKT, a.Nq, tid, Dt(t_begin & 1));
    }
    issue_stat(t_begin);
    if (loader) {
        ql.commit(Qt(t_begin & 1), tid);
        dl.commit(Dt(t_begin & 1), tid);
    }
    commit_stat(t_begin & 1);
    gen_lots(t_begin, t_begin & 1);
    ql.wait();
    __syncthreads();

    const int ts = drop_ts(a);
    f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[dt][i] = 0.f; dv[dt][i] = 0.f; }

    // per-lane LDS addresses: Q / dO row fragments share one set, their transposed fragments another; the tile buffer
    // is a compile-time constant of each instantiation of the step (the sweep is unrolled by two).
    const bf16* raddr[D / 16];
    const bf16* taddr[D / 32][2];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) raddr[s] = lds + row_frag_lane_off<D>(16 * s, lane);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
        int oa, ob;
        tr_frag_lane_off<D, true>(32 * dt, lane, oa, ob);
        taddr[dt][0] = lds + oa;
        taddr[dt][1] = lds + ob;
    }
    const float* stat_lane = stat + 4 * h;                            // row constants of query rows 4h + {0..3} (+ 8g + 32qt)
    const uint16_t* lots_lane = lots + 4 * h * LS + wave * LP + r;

    auto step = [&](auto buf_tag, int t) {
        constexpr int buf = decltype(buf_tag)::value;
        constexpr int QOFF = (buf * 2 + 0) * NS * TILE, DOFF = (buf * 2 + 1) * NS * TILE;
        if (t + 1 < nt) {
            if (loader) {
                ql.issue(qp, a.q_sn, (t + 1) * kKT, a.Nq, tid, Qt(buf ^ 1));
                dl.issue(dop, a.do_sn, (t + 1) * kKT, a.Nq, tid, Dt(buf ^ 1));
            }
            issue_stat(t + 1);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // the next tile's dropout lots are hashed between the two query slices: ahead of the second slice's MFMA chains
            // rather than in front of the barrier (dK/dV kernel -3 % same-box)
            if (qt == 1 && t + 1 < nt) gen_lots(t + 1, buf ^ 1);
            // S[q][key] = Q K^T ; dP[q][key] = dO V^T   (key on the lane)
            f32x16 s, dp, nd;
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // accumulator registers 4g .. 4g+3 <-> query rows 32 qt + 8 g + 4 h + {0..3}
                const int ro = 32 * qt + 8 * g;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(stat_lane + ((buf * 2 + 0) * kKT + ro));
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(stat_lane + ((buf * 2 + 1) * kKT + ro));
                // dropout: dS = Pd * dP_raw + P * (-delta), so the dP chain starts from zero and -delta is an ordinary operand
                // (one select per element instead of two, no second copy of the seed block)
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[4 * g + j] = l4[j]; nd[4 * g + j] = d4[j]; dp[4 * g + j] = DROP ? 0.f : d4[j]; }
            }
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
#pragma unroll
                for (int sa = 0; sa < NS; ++sa) {
                    bf16x8 qa = *reinterpret_cast<const bf16x8*>(raddr[ks] + (QOFF + sa * TILE + 32 * qt * D));
                    bf16x8 da = *reinterpret_cast<const bf16x8*>(raddr[ks] + (DOFF + sa * TILE + 32 * qt * D));
#pragma unroll
                    for (int sb = 0; sb < NS; ++sb)
                        if (sa + sb <= 1) {
                            s = mfma32(qa, kf[sb][ks], s);
                            dp = mfma32(da, vf[sb][ks], dp);
                        }
                }
            }
            float pd[16], ds[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);      // all exponentials ahead of the select / multiply pass (-1.3 % same-box)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ro = 32 * qt + 8 * g;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int i = 4 * g + j;
                    const f32x2 p2 = {s[i], s[i + 1]};
                    const f32x2 dp2 = {dp[i], dp[i + 1]};
                    f32x2 pd2 = p2, ds2;
                    if constexpr (DROP) {   // 1/(1-p) is folded into delta (pre-divided) and the epilogue scales
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const bool keep = (int16_t)lots_lane[(buf * kKT + ro + j + e) * LS] >= (int16_t)ts;
                            float pdv = keep ? p2[e] : 0.f;
                            asm("" : "+v"(pdv));     // select in fp32, so that the bf16 conversions below stay packed pairs
                            pd2[e] = pdv;
                        }
                        const f32x2 nd2 = {nd[i], nd[i + 1]};
                        ds2 = __builtin_elementwise_fma(pd2, dp2, p2 * nd2);      // packed multiply + packed fma
                    } else {
                        ds2 = p2 * dp2;                                           // packed multiply
                    }
                    pd[i] = pd2[0]; pd[i + 1] = pd2[1];
                    ds[i] = ds2[0]; ds[i + 1] = ds2[1];
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float x[8], y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { x[j] = pd[8 * s2 + j]; y[j] = ds[8 * s2 + j]; }
                bf16x8 pf[NS], dsf[NS];
                acc_split<NS>(x, pf);
                acc_split<NS>(y, dsf);
#pragma unroll
                for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
                    for (int sb = 0; sb < NS; ++sb) {
                        const int TO = sb * TILE + (32 * qt + 16 * s2) * D;
                        bf16x8 dof = tr_frag_at(taddr[dt][0] + (DOFF + TO), taddr[dt][1] + (DOFF + TO));
                        bf16x8 qf = tr_frag_at(taddr[dt][0] + (QOFF + TO), taddr[dt][1] + (QOFF + TO));
#pragma unroll
                        for (int sa = 0; sa < NS; ++sa)
                            if (sa + sb <= 1) {
                                dv[dt] = mfma32(pf[sa], dof, dv[dt]);
                                dk[dt] = mfma32(dsf[sa], qf, dk[dt]);
                            }
                    }
                }
            }
        }
        if (t + 1 < nt) {
            if (loader) {
                ql.commit(Qt(buf ^ 1), tid);
                dl.commit(Dt(buf ^ 1), tid);
            }
            commit_stat(buf ^ 1);
        }
        ql.wait();
        __syncthreads();
    };
    {
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        int t = t_begin;
        if (t < nt && (t & 1)) { step(B1{}, t); ++t; }            // align the sweep to an even tile
        for (; t + 1 < nt; t += 2) {
            step(B0{}, t);
            step(B1{}, t + 1);
        }
        if (t < nt) step(B0{}, t);
    }
    // tiles: rows = key (registers), col = d (lane)
    const float ks = DROP ? a.keep_scale : 1.f;
    if (a.qsplit > 1) {   // fp32 partial slabs [dK|dV][split][bh][key][d]; summed in a fixed order by attn_dkv_reduce_kernel
        const size_t slab = (size_t)a.B * a.H * a.Nk * D;
        float* pk = a.dkv_partial + ((size_t)split * a.B * a.H + bh) * a.Nk * D;
        float* pv = pk + (size_t)a.qsplit * slab;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = k0 + acc_row(i, h);
                if (kk < a.Nk) {
                    pk[(size_t)kk * D + 32 * dt + r] = dk[dt][i] * a.scale * ks;
                    pv[(size_t)kk * D + 32 * dt + r] = dv[dt][i] * ks;
                }
            }
        return;
    }
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = k0 + acc_row(i, h);
            if (kk < a.Nk) {
                dkp[(int64_t)kk * a.dk_sn + 32 * dt + r] = from_f<T>(dk[dt][i] * a.scale * ks);
                dvp[(int64_t)kk * a.dv_sn + 32 * dt + r] = from_f<T>(dv[dt][i] * ks);
            }
        }
}

// dK / dV = sum over the query-range slices of the partial slabs (fixed order), cast and scattered to the strided outputs
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(const AttnArgs a) {
    const size_t slab = (size_t)a.B * a.H * a.Nk * D;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < slab; e += (size_t)gridDim.x * 256) {
        const int d = (int)(e % D);
        const int key = (int)((e / D) % a.Nk);
        const int bh = (int)(e / ((size_t)D * a.Nk));
        const int b = bh / a.H, hh = bh % a.H;
        float sk = 0.f, sv = 0.f;
        for (int s = 0; s < a.qsplit; ++s) {
            sk += a.dkv_partial[(size_t)s * slab + e];
            sv += a.dkv_partial[((size_t)a.qsplit + s) * slab + e];
        }
        reinterpret_cast<T*>(a.dk)[b * a.dk_sb + hh * a.dk_sh + (int64_t)key * a.dk_sn + d] = from_f<T>(sk);
        reinterpret_cast<T*>(a.dv)[b * a.dv_sb + hh * a.dv_sh + (int64_t)key * a.dv_sn + d] = from_f<T>(sv);
    }
}

template <typename T, int D>
size_t fwd_lds_bytes() { return (size_t)2 * 2 * NSplit<T>::value * kKT * D * sizeof(bf16); }
template <typename T, int D>
size_t dkv_lds_bytes(bool drop, int waves = 4) { return fwd_lds_bytes<T, D>() + 2 * 2 * kKT * sizeof(float) + (drop ? 2 * kKT * (waves == 4 ? LotTile<4>::ROW : LotTile<8>::ROW) * sizeof(uint16_t) : 0); }

// Raises a kernel's dynamic-LDS limit once per (kernel, size): the attribute is sticky, and a driver call per launch would
// also sit inside hipGraph captures of the training step.
template <typename K>
hipError_t set_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    static thread_local const void* done_fn[64];
    static thread_local size_t done_bytes[64];
    static thread_local int ndone = 0;
    const void* fn = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < ndone; ++i)
        if (done_fn[i] == fn && done_bytes[i] >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && ndone < 64) { done_fn[ndone] = fn; done_bytes[ndone] = bytes; ++ndone; }
    return e;
}

// Experiments only (scripts/attn_ab.py): HVC_ATTN_EXTRA_LDS=<bytes> pads every attention launch's dynamic LDS request, e.g. to
// force one workgroup per CU and read off how much two co-resident waves per SIMD overlap (profiles/r03_attention_*).
size_t extra_lds() {
    return (size_t)option(kOptAttnExtraLds);
}

template <typename T, int D, bool DROP, bool VEC>
hipError_t launch_fwd(const AttnArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && VEC && HVC_ATTN_DMA) {
        // software-pipelined kernel (256-row workgroups of eight wavefronts, one per CU) once they fill every CU twice over;
        // HVC_ATTN_PIPE=0 keeps the phase-separated kernels, =2 takes the pipelined one on any grid (tests); a row pin (below) wins
        const int pipe = option(kOptAttnPipe);
        const int64_t nwgp = (int64_t)((a.Nq + 255) / 256) * a.B * a.H;
        // its tile loader addresses K / V rows of one (b, h) by 32-bit byte offsets from a wave-uniform base
        const bool span_ok = ((int64_t)(a.Nk - 1) * a.k_sn + D) * 2 < (int64_t(1) << 31) && ((int64_t)(a.Nk - 1) * a.v_sn + D) * 2 < (int64_t(1) << 31);
        if (option(kOptAttnFwdRows) == 0 && span_ok && (pipe == 2 || (pipe == 1 && nwgp >= 512))) {
            const size_t ldsp = (size_t)(kKBufs + 4) * kKT * D * sizeof(bf16) + extra_lds();
            // d = 32: 128-row workgroups of four wavefronts, two or three per CU (measured -6 % against the 8-wavefront form there:
            // half the vector work per MFMA hides under fewer co-resident barriers); d = 64: eight wavefronts.  HVC_ATTN_FWD_WAVES pins.
            const int wpin = option(kOptAttnFwdWaves);
            if (wpin == 4 || (wpin != 8 && D == 32)) {
                auto kp4 = attn_fwdp_kernel<D, DROP, 4>;
                hipError_t e4 = set_lds(kp4, ldsp);
                if (e4 != hipSuccess) return e4;
                hipLaunchKernelGGL(kp4, dim3((unsigned)(((a.Nq + 127) / 128) * a.B * a.H)), dim3(256), ldsp, st, a);
                return hipGetLastError();
            }
            auto kp8 = attn_fwdp_kernel<D, DROP, 8>;
            hipError_t ep = set_lds(kp8, ldsp);
            if (ep != hipSuccess) return ep;
            hipLaunchKernelGGL(kp8, dim3((unsigned)nwgp), dim3(512), ldsp, st, a);
            return hipGetLastError();
        }
    }
    if constexpr (sizeof(T) == 2 && VEC) {
        // 64 rows per wavefront once its 256-row workgroups fill every CU twice over; smaller problems keep the 128-row
        // workgroups (more of them, three per CU)
        const int nqb2 = (a.Nq + kQB2 - 1) / kQB2;
        const int force = option(kOptAttnFwdRows);          // 64 / 32: pin the kernel (parity tests run both on every shape)
        if (force == 64 || (force != 32 && nqb2 * a.B * a.H >= 512)) {
            const size_t lds2 = (size_t)2 * 2 * kKT2 * D * sizeof(bf16) + extra_lds();
            // d = 64: 512-row workgroups of eight wavefronts (one per CU) once they fill every CU twice over; HVC_ATTN_FWD_WAVES=4|8 pins the form
            if constexpr (D == 64) {
                const int wpin = option(kOptAttnFwdWaves);
                if (wpin != 4 && (wpin == 8 || (int64_t)((a.Nq + 511) / 512) * a.B * a.H >= 512)) {
                    auto k8 = attn_fwd2_kernel<D, DROP, 8>;
                    hipError_t e8 = set_lds(k8, lds2);
                    if (e8 != hipSuccess) return e8;
                    hipLaunchKernelGGL(k8, dim3(((a.Nq + 511) / 512) * a.B * a.H), dim3(512), lds2, st, a);
                    return hipGetLastError();
                }
            }
            auto k2 = attn_fwd2_kernel<D, DROP>;
            hipError_t e2 = set_lds(k2, lds2);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(k2, dim3(nqb2 * a.B * a.H), dim3(256), lds2, st, a);
            return hipGetLastError();
        }
    }
    const int nqb = (a.Nq + kQB - 1) / kQB;
    const size_t lds = fwd_lds_bytes<T, D>() + extra_lds();
    auto k = attn_fwd_kernel<T, D, DROP, VEC>;
    hipError_t e = set_lds(k, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(nqb * a.B * a.H), dim3(256), lds, st, a);
    return hipGetLastError();
}

// HVC_ATTN_BWD_WAVES = 4 | 8: pin the workgroup form of the dQ and dK/dV kernels (A/B timing, parity tests of both forms); 0 = by size
inline int bwd_waves_pin() {
    return option(kOptAttnBwdWaves);
}

template <typename T, int D, bool DROP, bool VEC>
hipError_t launch_bwd(const AttnArgs& a, hipStream_t st) {
    const int ph = a.phases ? a.phases : 7;
    if (ph & 1) {
        const int64_t nthreads = (int64_t)a.B * a.H * a.Nq * (D / 8);
        hipLaunchKernelGGL((attn_delta_kernel<T, D>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (ph & 2) {
        // 256-key workgroups of eight wavefronts once they fill every CU twice over (HVC_ATTN_BWD_WAVES=4 / 8 pins the form)
        constexpr bool CAN8 = sizeof(T) == 2 && VEC;
        const int pin = bwd_waves_pin();
        // (d = 32: the four-wavefront form, three workgroups per CU - see LotTile - unless pinned; few key blocks with the query range
        // sliced (cross-attention): eight wavefronts, one workgroup per CU: -4.5 % same-box at d = 64 and d = 32)
        const int qs_try = attention_bwd_qsplit(a.B, a.H, a.Nq, a.Nk);
        const bool sliced = qs_try > 1 && a.dkv_partial && a.partial_floats >= (int64_t)2 * qs_try * a.B * a.H * a.Nk * D;
        const bool w8 = CAN8 && pin != 4 && (pin == 8 || sliced || (D != 32 && (int64_t)((a.Nk + 255) / 256) * a.B * a.H >= 512));
        const int KBh = w8 ? 256 : kQB;
        const int nkb = (a.Nk + KBh - 1) / KBh;
        const size_t lds = dkv_lds_bytes<T, D>(DROP, w8 ? 8 : 4) + extra_lds();
        // Few key blocks (cross-attention, small contexts) leave most CUs idle: slice the query range over more
        // workgroups and sum the fp32 partial dK / dV slabs in a second, deterministic pass.
        AttnArgs b = a;
        b.qsplit = attention_bwd_qsplit(a.B, a.H, a.Nq, a.Nk);
        if (b.qsplit > 1 && (!a.dkv_partial || a.partial_floats < (int64_t)2 * b.qsplit * a.B * a.H * a.Nk * D)) b.qsplit = 1;
        hipError_t e = hipSuccess;
        if constexpr (CAN8) {
            if (w8) {
                auto k8 = attn_bwd_dkv_kernel<T, D, DROP, VEC, 8>;
                e = set_lds(k8, lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k8, dim3(nkb * a.B * a.H * b.qsplit), dim3(512), lds, st, b);
            }
        }
        if (!w8) {
            auto k = attn_bwd_dkv_kernel<T, D, DROP, VEC>;
            e = set_lds(k, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(nkb * a.B * a.H * b.qsplit), dim3(256), lds, st, b);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (b.qsplit > 1) {
            size_t n = (size_t)a.B * a.H * a.Nk * D;
            int blocks = (int)((n + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL((attn_dkv_reduce_kernel<T, D>), dim3(blocks), dim3(256), 0, st, b);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    if (ph & 4) {
        const size_t lds = fwd_lds_bytes<T, D>() + extra_lds();
        if constexpr (sizeof(T) == 2 && VEC) {
            const int pin = bwd_waves_pin();
            // (d = 32: the four-wavefront form at <= 168 registers, three workgroups per CU: -2.8 % same-box against the eight-wavefront form)
            if (pin != 4 && (pin == 8 || (D != 32 && (int64_t)((a.Nq + 255) / 256) * a.B * a.H >= 512))) {
                const int nqb = (a.Nq + 255) / 256;
                auto k8 = attn_bwd_dq_kernel<T, D, DROP, VEC, 8>;
                hipError_t e = set_lds(k8, lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k8, dim3(nqb * a.B * a.H), dim3(512), lds, st, a);
                return hipGetLastError();
            }
        }
        const int nqb = (a.Nq + kQB - 1) / kQB;
        auto k = attn_bwd_dq_kernel<T, D, DROP, VEC>;
        hipError_t e = set_lds(k, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(nqb * a.B * a.H), dim3(256), lds, st, a);
        return hipGetLastError();
    }
    return hipSuccess;
}

template <typename T, int D, bool DROP>
hipError_t dispatch_vec(const AttnArgs& a, bool bwd, hipStream_t st) {
    if (a.vec) return bwd ? launch_bwd<T, D, DROP, true>(a, st) : launch_fwd<T, D, DROP, true>(a, st);
    return bwd ? launch_bwd<T, D, DROP, false>(a, st) : launch_fwd<T, D, DROP, false>(a, st);
}

template <typename T>
hipError_t dispatch(const AttnArgs& a, bool bwd, hipStream_t st) {
    const bool drop = a.drop_thresh != 0;
    if (a.D == 64) return drop ? dispatch_vec<T, 64, true>(a, bwd, st) : dispatch_vec<T, 64, false>(a, bwd, st);
    if (a.D == 32) return drop ? dispatch_vec<T, 32, true>(a, bwd, st) : dispatch_vec<T, 32, false>(a, bwd, st);
    return hipErrorInvalidValue;
}

}  // namespace

int attention_bwd_qsplit(int B, int H, int Nq, int Nk) {
    const int wgs = ((Nk + kQB - 1) / kQB) * B * H;
    const int nt = (Nq + kKT - 1) / kKT;
    if (wgs >= 512 || nt < 16) return 1;
    int want = (512 + wgs - 1) / wgs;          // the kernel runs two workgroups per CU: fill 512 slots in whole rounds (768 left half a round idle)
    if (want > nt / 8) want = nt / 8;          // >= 8 query tiles per slice
    return want < 2 ? 1 : want;
}

hipError_t attention_launch(const AttnArgs& a, bool bwd, hipStream_t st) {
    return a.is_bf16 ? dispatch<bf16>(a, bwd, st) : dispatch<float>(a, bwd, st);
}

}  // namespace hvc
