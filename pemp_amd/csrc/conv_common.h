// Shared between the conv engine's translation units.
#pragma once
#include "common.h"

namespace pemp {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float v4f;   // first-class vector: loads/stores never become memcpy
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// internal flag (not part of pemp_hip.h's desc flags): y and the residual are bf16 tensors (strides in bf16 elements)
#define PEMP_CONV_BF16_IO 0x100u

// four consecutive channels at element offset ``off`` of a tensor that is fp32 or (bf16 != 0) bf16, as fp32
__device__ __forceinline__ v4f load_quad(const float* base, size_t off, unsigned bf16) {
    if (!bf16) return *(const v4f*)(base + off);
    const u32x2 r = *(const u32x2*)((const unsigned short*)base + off);
    return v4f{__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xFFFF0000u),
               __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xFFFF0000u)};
}

// fp32 -> bf16, round to nearest even (NaN stays NaN: the quiet bit is forced)
__device__ __forceinline__ unsigned int bf16_bits(float f) {
    const unsigned int u = __builtin_bit_cast(unsigned int, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ void store_quad(float* base, size_t off, const v4f& o, unsigned bf16) {
    if (!bf16) {
        *(v4f*)(base + off) = o;
        return;
    }
    u32x2 r;
    r.x = bf16_bits(o.x) | (bf16_bits(o.y) << 16);
    r.y = bf16_bits(o.z) | (bf16_bits(o.w) << 16);
    *(u32x2*)((unsigned short*)base + off) = r;
}

struct ConvArgs {
    const float* x;
    const float* w;
    float* y;
    const float* scale;
    const float* shift;
    const float* res;
    const float* padv;      // per-input-channel value of out-of-image taps (NULL: zero padding)
    float* stats;           // [ceil(M/BM)][2][Cout] per-row-tile partial sums (NULL: none; conv_dma2.hip only): of y and y^2, or,
                            // with bz set, of g and g*xhat (BatchNorm backward; see pemp_conv2d_bnbwd_nhwc_f32)
    const uint32_t* bmask;  // sign bits of the BatchNorm's output (NULL: no ReLU)
    const float* bz;        // the BatchNorm's input, per-pixel stride ldbz
    const float* bmean;
    const float* binvstd;
    int ldbz;
    // DropBlock2D behind the conv (pemp_conv2d_dropblock_nhwc_f32): every output row m is multiplied by rowmask[m] * M / *rowcnt
    // (the layer's two statements, in its order and rounding: ((y * mask) * numel) / sum(mask)); NULL: none
    const float* rowmask;
    const int* rowcnt;
    // split-K of the remainder tiles (conv_dma2.hip, SK kernels): tiles >= sk_full are computed by sk_S blocks each, every block
    // over 1/sk_S of the K steps; partial accumulators go to sk_ws, the last block to arrive adds them in piece order
    float* sk_ws;
    int* sk_cnt;            // one arrival counter per split tile: zero on entry, left zero
    int sk_full, sk_S;
    int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, KH, KW, stride, pad, dil, ldr, Kpad;
    unsigned flags;
    int M, HoWo, cin_steps, nk, ntaps;
    int bm_first = 0;       // conv_dma2.hip: the launch's first row tile (in units of its BM; 0 except in the hybrid launch's second member)
};

// up to CONV_GROUP_MAX independent convs of one tile shape in one launch (conv_dma2.hip: conv_dma2_group_kernel)
constexpr int CONV_GROUP_MAX = 4;
struct ConvGroupArgs {
    ConvArgs a[CONV_GROUP_MAX];
    int first[CONV_GROUP_MAX + 1];   // first block of member i (multiples of 8); first[n] = grid size
    int nblk[CONV_GROUP_MAX];        // tiles of member i
    int n;
};

// XCD-aware tile order.  Hardware deals consecutive block ids round-robin over the 8 XCDs (each
// with a private 4 MiB L2).  Tiles that share an A row-panel (same M tile, different N tiles) and
// neighbouring M tiles (3x3 halos) should therefore get ids that are congruent mod 8.  This
// bijection hands every XCD one contiguous range of the logical (M-major, N-minor) tile order.
// Placement only affects speed, never results.
__device__ __forceinline__ int xcd_tile_order(int bid, int nblk) {
    const int x = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, rem = nblk & 7;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + idx;
}

// Wave-level epilogue through LDS.  The 32x32 accumulator tile sits "column per lane" (lane = channel,
// 16 registers = pixels), which would mean 16 scalar residual loads + 16 scalar stores per lane.  Each
// wave transposes its tile through a private 4 KB LDS patch so that a lane owns 4 consecutive channels
// of one pixel: float4 residual loads and float4 stores, 8 lanes = one 128-B row segment.  Same
// arithmetic, in the same order, as the scalar form:  v = acc * scale + (shift (+ per-image) (+ res)).
// ``pre`` (optional): the residual quads of the wave's tiles, loaded by the caller ahead of time in the order
// [mi][ni][i] (row = (lane >> 3) + 8 i of tile (mi, ni), channels (lane & 7) * 4 ..) -- conv_dma2.hip issues those loads
// under its last K step so that their latency is not exposed here.
// EPI 1: per-channel sums of the stored values and of their squares over the wave's rows, left in R ([TN][2][8][32] floats of
// LDS per wave: one row of 32 channel sums per lane group; the kernel adds the lane groups and the waves of a block in a fixed
// order and writes one partial row per row tile: the batch statistics of the BatchNorm behind the conv; fixed layout, fixed
// order -> deterministic).
// EPI 2: the stored value is g = o masked by the sign bits of the BatchNorm output this gradient belongs to, the sums are
// those of g and g * xhat (first half of that BatchNorm's backward; the expression of colsum_kernel<1> in train_ops.hip).
// DB: DropBlock2D's scaling of the output rows (a.rowmask / a.rowcnt) -- a compile-time variant, instantiated for the kernels of
// pemp_conv2d_dropblock_nhwc_f32 only: it keeps the four instructions out of every other epilogue.  (Round 4 made it one because
// the run-time branch "broke" one kernel; the cause was an unrelated wait-state hazard in conv_dma2.hip's inline asm that the
// branch merely re-scheduled into view -- DESIGN.md section 4, scratch/t31/README.md.)
template <int TM, int TN, int NPRE, int EPI = 0, bool DB = false>
__device__ __forceinline__ void conv_epilogue_lds_pre(const ConvArgs& a, f32x16 (&acc)[TM][TN], float* S, int m_base,
                                                      int n_base, int lane, const v4f (&pre)[NPRE], float* R = nullptr) {
    constexpr bool PRE = NPRE == TM * TN * 4;       // (an array of 1 = "no prefetched residual": registers, never scratch)
    const bool relu = a.flags & PEMP_CONV_RELU;
    const bool per_img = a.flags & PEMP_CONV_SHIFT_PER_IMAGE;
    float db_sum = 1.f, db_numel = 1.f;
    if constexpr (DB) {
        db_sum = (float)*a.rowcnt;
        db_numel = (float)a.M;
    }
    const int lr = lane & 31, lh = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n_base + ni * 32 + c4;
        v4f sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) sc = *(const v4f*)(a.scale + n);
        if (a.shift && !per_img) sh = *(const v4f*)(a.shift + n);
        v4f bmu = {0.f, 0.f, 0.f, 0.f}, bis = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == 2) {
            bmu = *(const v4f*)(a.bmean + n);
            bis = *(const v4f*)(a.binvstd + n);
        }
        v4f t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};      // EPI: sums over the wave's TM row groups, ascending
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            v4f zt[4];
            uint32_t mw[4];
            if constexpr (EPI == 2) {                              // issued before the accumulators go through LDS
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m_base + mi * 32 + rr + 8 * i;
                    zt[i] = v4f{0.f, 0.f, 0.f, 0.f};
                    mw[i] = 0xFFFFFFFFu;
                    if (m < a.M) {
                        zt[i] = *(const v4f*)(a.bz + (size_t)m * a.ldbz + n);
                        if (a.bmask) mw[i] = a.bmask[(size_t)m * (a.Cout >> 5) + (n >> 5)];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) S[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + lr] = acc[mi][ni][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's writes have landed (DS is in-order per wave)
            v4f s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + 8 * i;
                const int m = m_base + mi * 32 + row;
                const v4f v = *(const v4f*)(S + row * 32 + c4);
                if (m < a.M) {
                    v4f add = sh;
                    if (per_img) add += *(const v4f*)(a.shift + (size_t)(m / a.HoWo) * a.Cout + n);
                    if (a.res) {
                        if constexpr (PRE) add += pre[(mi * TN + ni) * 4 + i];
                        else add += load_quad(a.res, (size_t)m * a.ldr + n, a.flags & PEMP_CONV_BF16_IO);
                    }
                    v4f o;
                    o.x = __builtin_fmaf(v.x, sc.x, add.x);      // explicit: every epilogue variant must round identically
                    o.y = __builtin_fmaf(v.y, sc.y, add.y);
                    o.z = __builtin_fmaf(v.z, sc.z, add.z);
                    o.w = __builtin_fmaf(v.w, sc.w, add.w);
                    if (relu) {
                        o.x = fmaxf(o.x, 0.f);
                        o.y = fmaxf(o.y, 0.f);
                        o.z = fmaxf(o.z, 0.f);
                        o.w = fmaxf(o.w, 0.f);
                    }
                    if constexpr (DB) {                           // DropBlock2D.forward / its backward, fused (dropout.hip: pixel_scale_kernel)
                        const float k = a.rowmask[m];
                        o.x = __fdiv_rn(__fmul_rn(__fmul_rn(o.x, k), db_numel), db_sum);
                        o.y = __fdiv_rn(__fmul_rn(__fmul_rn(o.y, k), db_numel), db_sum);
                        o.z = __fdiv_rn(__fmul_rn(__fmul_rn(o.z, k), db_numel), db_sum);
                        o.w = __fdiv_rn(__fmul_rn(__fmul_rn(o.w, k), db_numel), db_sum);
                    }
                    if constexpr (EPI == 2) {
                        const uint32_t b = mw[i] >> c4;
                        o.x = (b & 1u) ? o.x : 0.f;
                        o.y = (b & 2u) ? o.y : 0.f;
                        o.z = (b & 4u) ? o.z : 0.f;
                        o.w = (b & 8u) ? o.w : 0.f;
                    }
                    store_quad(a.y, (size_t)m * a.ldy + n, o, a.flags & PEMP_CONV_BF16_IO);
                    if constexpr (EPI == 1) {
                        s1 += o;
                        s2 += o * o;
                    }
                    if constexpr (EPI == 2) {
                        s1 += o;
                        s2 += o * ((zt[i] - bmu) * bis);
                    }
                }
            }
            if constexpr (EPI != 0) {       // per lane: its 4 rows of this row group, then the wave's row groups in ascending order
                t1 += s1;
                t2 += s2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the patch is rewritten
        }
        if constexpr (EPI != 0) {
            // The 8 lanes that share a channel quad (lane bits 3..5 = the row inside an 8-row group) leave their sums side by side
            // in LDS; conv_stats_store adds them in a fixed order.  (Round 3 reduced them here with a three-level shuffle butterfly
            // per 32 x 32 sub-tile: 24 cross-lane operations each, 20 us of a 105 us launch on the 256 -> 1024 layers.)
            *(v4f*)(R + ((ni * 2 + 0) * 8 + rr) * 32 + c4) = t1;
            *(v4f*)(R + ((ni * 2 + 1) * 8 + rr) * 32 + c4) = t2;
        }
    }
}

// Block-level finish of the EPI sums: the waves that cover the same columns (WGM of them, one per row band of the tile) are
// added in ascending row order and the tile's partial row goes to a.stats[bm].  Rall: the R areas of all waves
// ([NW][TN][2][8][32]); call after a block barrier.
template <int BN, int WGM, int NW, int TN>
__device__ __forceinline__ void conv_stats_store(const ConvArgs& a, const float* Rall, int bm, int n0, int tid) {
    constexpr int WGN = NW / WGM, WN = BN / WGN;
    if (tid < 2 * BN) {
        const int st = tid / BN, col = tid - st * BN;
        const int wni = col / WN, ni = (col - wni * WN) >> 5, c = col & 31;
        float s = 0.f;
#pragma unroll
        for (int wmi = 0; wmi < WGM; ++wmi)
#pragma unroll
            for (int g = 0; g < 8; ++g) s += Rall[((((wmi * WGN + wni) * TN + ni) * 2 + st) * 8 + g) * 32 + c];
        a.stats[((size_t)bm * 2 + st) * a.Cout + n0 + col] = s;
    }
}

template <int TM, int TN, bool DB = false>
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& a, f32x16 (&acc)[TM][TN], float* S, int m_base,
                                                  int n_base, int lane) {
    const v4f none[1] = {{0.f, 0.f, 0.f, 0.f}};
    static_assert(TM * TN * 4 != 1, "tile");
    conv_epilogue_lds_pre<TM, TN, 1, 0, DB>(a, acc, S, m_base, n_base, lane, none);
}

// Epilogue of the 16-row wave tiles (conv_dma2.hip, R16): TN accumulators of 16 x 16 (rows 4 (lane >> 4) + e, column lane & 15) go
// through a 16 x (16 TN) LDS patch (row stride + 4 floats: the four lane groups write different banks) so that a lane owns 4
// consecutive channels of one pixel; the arithmetic is conv_epilogue_lds_pre's, statement for statement (identical rounding).
template <int TN>
__device__ __forceinline__ void conv_epilogue_r16(const ConvArgs& a, v4f (&acc)[TN], float* S, int m_base, int n_base, int lane) {
    constexpr int WN = TN * 16, LD = WN + 4, QPR = WN / 4;
    static_assert(16 * LD * 4 <= 4096 && (16 * QPR) % 64 == 0, "R16 epilogue patch");
    const bool relu = a.flags & PEMP_CONV_RELU;
    const bool per_img = a.flags & PEMP_CONV_SHIFT_PER_IMAGE;
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(4 * g + e) * LD + ni * 16 + r16] = acc[ni][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's writes have landed (DS is in-order per wave)
    const int c4 = (lane % QPR) * 4, n = n_base + c4;
    v4f sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.scale) sc = *(const v4f*)(a.scale + n);
    if (a.shift && !per_img) sh = *(const v4f*)(a.shift + n);
#pragma unroll
    for (int i = 0; i < 16 * QPR / 64; ++i) {
        const int row = (lane + 64 * i) / QPR;
        const int m = m_base + row;
        const v4f v = *(const v4f*)(S + row * LD + c4);
        if (m < a.M) {
            v4f add = sh;
            if (per_img) add += *(const v4f*)(a.shift + (size_t)(m / a.HoWo) * a.Cout + n);
            if (a.res) add += load_quad(a.res, (size_t)m * a.ldr + n, a.flags & PEMP_CONV_BF16_IO);
            v4f o;
            o.x = __builtin_fmaf(v.x, sc.x, add.x);
            o.y = __builtin_fmaf(v.y, sc.y, add.y);
            o.z = __builtin_fmaf(v.z, sc.z, add.z);
            o.w = __builtin_fmaf(v.w, sc.w, add.w);
            if (relu) {
                o.x = fmaxf(o.x, 0.f);
                o.y = fmaxf(o.y, 0.f);
                o.z = fmaxf(o.z, 0.f);
                o.w = fmaxf(o.w, 0.f);
            }
            store_quad(a.y, (size_t)m * a.ldy + n, o, a.flags & PEMP_CONV_BF16_IO);
        }
    }
}

// conv_dma.hip
int launch_conv_dma(int tile, const ConvArgs& a, hipStream_t st);
// conv_dma2.hip
bool conv_dma2_supported(const ConvArgs& a);
int launch_conv_dma2(int tile, const ConvArgs& a, hipStream_t st);
int launch_conv_dma2_group(int tile, ConvGroupArgs& g, hipStream_t st);      // fills g.first / g.nblk
int launch_conv_dma2_hybrid(const ConvArgs& a, hipStream_t st);             // tile id 29; -2: the geometry has no hybrid split
int conv_dma2_hybrid_rows(int M, int Cout);                               // rows on the 64 x 64 tile in that launch (0: no split)
int launch_conv_dma2_bf16(int tile, const ConvArgs& a, hipStream_t st);      // bf16 operands (Cin / ldx / Kpad in dwords)
int launch_conv_dma2_db(int tile, ConvArgs a, void* ws, size_t ws_bytes, bool split, hipStream_t st);   // + DropBlock row scaling
int conv_dma2_tile_rows(int tile);
// split-K plan of tile variant `tile` (1..7) for this geometry: number of unsplit tiles, split tiles, pieces per split tile
// (pieces == 1: the variant runs unsplit) and the workspace the launch needs (counters first, then the partial tiles)
struct SplitKPlan { int full, split, pieces; size_t ws_bytes; };
SplitKPlan conv_dma2_splitk_plan(int tile, const ConvArgs& a);
int launch_conv_dma2_splitk(int tile, ConvArgs a, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace pemp
