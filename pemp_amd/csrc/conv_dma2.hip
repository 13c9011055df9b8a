// Implicit-GEMM convolution, second LDS-DMA variant: the same GEMM view, LDS image, fragment reads, MFMA order and
// epilogue as conv_dma.hip (results are bit-identical), with the two things that kept the MFMA pipe idle there removed:
//
//  * Operand addressing.  conv_dma.hip rebuilds a 64-bit source pointer per row every K step (bounds tests, a 64-bit
//    multiply-add, a select against the zero block: ~100 VALU/SALU instructions per wave per step, executed by both
//    waves of a SIMD at the same moment right after the barrier).  Here both operands are fetched with
//    `buffer_load_dwordx4 ... offen lds`: a per-lane BYTE OFFSET that never changes during the K loop plus a
//    wave-uniform SGPR offset that carries the whole K-step dependence (tap displacement + channel chunk for the
//    activations, K offset for the weights).  Which taps of a row fall outside the image is decided ONCE per tile
//    (one bit per tap); an out-of-image tap replaces the lane's offset by 2^31, the buffer range check (num_records =
//    2 GiB) then makes the hardware write zeros -- zero padding without a zero block, 3 VALU instructions per row
//    and step.  (The base of the activation descriptor is moved back by pad rows + pad pixels so that every
//    in-image offset is non-negative.)
//
//  * Barrier placement.  One barrier per K step as before, but it sits BEFORE the last quarter of the step's MFMAs,
//    whose operands are already in registers: after the barrier a wave issues the first fragment reads of the next
//    step and the LDS-DMA of the step after that in the shadow of those 4*TM*TN MFMAs, instead of starting every
//    step with address arithmetic + a dozen LDS reads + a full wait while the matrix pipe drains.
//
// Not handled here (conv_dma.hip keeps them): the NHWC4 stem (a K step spans 8 taps, the tap differs per lane), more
// than 32 taps, operands of 2 GiB or more, and a per-channel padding VALUE (padv) that does not lie behind the
// activations in the same 2 GiB window: the in-image and the out-of-image lanes of a piece must come through ONE
// descriptor (an exec-masked LDS-DMA does not leave the inactive lanes' 16-byte slots alone -- tried: two masked DMAs per
// piece give wrong data -- so a piece cannot be assembled from two).
#include "conv_common.h"

#ifndef PEMP_SK_ACQUIRE
#define PEMP_SK_ACQUIRE 1     // 0: round 4's hand-off without the consumer acquire (A/B builds only)
#endif

namespace pemp {

typedef __attribute__((address_space(3))) void* lptr_t;

// The block's work: tile ``bid`` of the ``nblk`` blocks of one conv (the plain kernel passes blockIdx.x / gridDim.x; the
// grouped kernel below the block's index inside its member conv).
// BF16 (the side-figure variant, pemp_conv2d_bf16_nhwc): the operands are bf16 in memory.  The kernel is the same down to the
// byte -- a K step is still 128 bytes per row, staged, swizzled and read as 16-byte quads -- because the caller hands over
// Cin / ldx / Kpad in DWORDS (two bf16 each): a quad then holds 8 bf16 = the 8 K values one lane feeds into
// v_mfma_f32_32x32x16_bf16 (lanes 0-31: K 0..7, lanes 32-63: K 8..15 = quads 2j and 2j + 1, exactly the pair a fragment read
// of step j fetches), so one MFMA does the work of the four v_mfma_f32_32x32x2_f32 of the fp32 kernel at 8 cycles instead of
// 4 x 64.  Accumulation stays fp32; the epilogue writes bf16 (or fp32 for the last layer) -- PEMP_CONV_BF16_IO.
// R16 (tile id 28: 32 x 64 block, four waves of 16 x 32): the same K loop on v_mfma_f32_16x16x4_f32.  Measured on gfx950
// (scratch/mfma_eq): the fp32 MFMAs of every shape accumulate as ONE sequential fma chain in their k order -- 32x32x2, 16x16x4
// and fmaf() agree bit for bit -- so a 16-row wave tile that feeds the chain in the order of the 32-row kernels (per quarter j
// of a K step: k = 8j + {0, 4, 1, 5, 2, 6, 3, 7}) is BIT-IDENTICAL to them and may serve the exact evaluation path.  What it
// buys: granularity.  A one-episode step has 5202 output rows; a 256-channel conv is 1304 wave tiles of 32 x 32 on 1024 SIMDs --
// two rounds, the second 27 % full, 64 % of the chip's MFMA time at best -- and 2608 of 16 x 32 -- three rounds of half the
// length, 85 %.  It pays twice the LDS reads per flop for that, so it only wins where a launch is a few rounds long.
template <int BM, int BN, int WGM, int NW, bool PADV, int EPI = 0, bool SK = false, bool BF16 = false, bool DB = false, bool R16 = false>
__device__ __forceinline__ void conv_dma2_body(const ConvArgs& a, const int bid, const int nblk) {
#if defined(__HIP_DEVICE_COMPILE__)     // the host pass only needs the launch stub (buffer-resource builtins / "s" asm operands are device-only)
    constexpr int WGN = NW / WGM;
    constexpr int RPI = NW * 8;                 // rows covered by one DMA instruction round of the block
    constexpr int WM = BM / WGM, WN = BN / WGN;
    static_assert(!R16 || (WM == 16 && WN % 16 == 0 && EPI == 0 && !SK && !BF16 && !DB), "R16: 16-row wave tiles, plain epilogue only");
    constexpr int TM = R16 ? 1 : WM / 32, TN = R16 ? WN / 16 : WN / 32;      // R16: TN counts 16-column MFMA tiles
    constexpr int AL = BM / RPI, BL = BN / RPI; // DMA wave-instructions per thread per K step
    constexpr int NMF = R16 ? 2 * TN : TM * TN * (BF16 ? 1 : 4), NDS = R16 ? 1 + TN : TM + TN, NDMA = AL + BL;   // per quarter step: MFMAs, fragment reads; DMAs per step
    constexpr int PER = (NDS + NDMA + NMF - 1) / NMF;

    extern __shared__ __attribute__((aligned(16))) v4f smem[];
    v4f* As = smem;                      // [2][BM][8]
    v4f* Bs = smem + 2 * BM * 8;         // [2][BN][8]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WGN) * WM;
    const int wn0 = (wave % WGN) * WN;
    const int lr = lane & 31, lh = lane >> 5;

    const int ntn = a.Cout / BN;
    // SK: blocks [0, sk_full) compute whole tiles; behind them, sk_S consecutive blocks share one of the remaining tiles,
    // block `piece` of them running K steps [kt0, kt0 + nkl)
    int tile_id, kt0 = 0, nkl = a.nk, sk_r = -1, piece = 0;
    if constexpr (SK) {
        const int b = bid;
        if (b < a.sk_full) {
            tile_id = xcd_tile_order(b, a.sk_full);
        } else {
            const int rb = b - a.sk_full;
            sk_r = rb / a.sk_S;
            piece = rb - sk_r * a.sk_S;
            tile_id = a.sk_full + sk_r;
            kt0 = (int)((long long)piece * a.nk / a.sk_S);
            nkl = (int)((long long)(piece + 1) * a.nk / a.sk_S) - kt0;
        }
    } else {
        tile_id = xcd_tile_order(bid, nblk);
    }
    const int bm = a.bm_first + tile_id / ntn;
    const int bn = tile_id % ntn;
    const int m0 = bm * BM, n0 = bn * BN;

    // loader role: thread (r, p) fetches, for rows r + RPI i, the quad that belongs at position p (see conv_dma.hip)
    const int p = tid & 7;
    const int r = tid >> 3;
    const int sq = p ^ ((r >> 1) & 7);

    const int bias_pix = a.pad * a.W + a.pad;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - (ptrdiff_t)bias_pix * a.ldx), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x80000000u, 0x00020000);

    // PADV (ops.fold_input_affine: a BatchNorm in front of a zero-padded conv): out-of-image taps read a per-channel VALUE
    // instead of zero.  One descriptor must serve both kinds of lane, so this variant needs the [Cin] vector INSIDE the
    // activation descriptor's range, behind the tensor (the engine allocates it there); an out-of-image lane then gets
    // the offset of its channel quad of that vector, minus the tap displacement the SGPR offset is about to add.
    const unsigned padv_off = PADV ? (unsigned)((const char*)a.padv - (const char*)(a.x - (ptrdiff_t)bias_pix * a.ldx)) + sq * 16 : 0x80000000u;
    unsigned a_voff[AL], a_inv[AL], b_voff[BL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + r + RPI * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int img = mm / a.HoWo;
        const int rem = mm - img * a.HoWo;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        const int hi0 = ho * a.stride - a.pad;
        const int wi0 = wo * a.stride - a.pad;
        a_voff[i] = (unsigned)(((img * a.H + hi0) * a.W + wi0 + bias_pix) * a.ldx + sq * 4) * 4u;
        unsigned mask = 0;
        int kh = 0, kw = 0;
        for (int t = 0; t < a.ntaps; ++t) {
            const int hi = hi0 + kh * a.dil, wi = wi0 + kw * a.dil;
            if (ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) mask |= 1u << t;
            if (++kw == a.KW) {
                kw = 0;
                ++kh;
            }
        }
        a_inv[i] = ~mask;
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) b_voff[i] = (unsigned)((n0 + r + RPI * i) * a.Kpad + sq * 4) * 4u;

    // wave-uniform K-step state.  Multi-tap convs: channel chunk OUTER, tap INNER (same order as every other conv
    // kernel of the library: variants stay bit-identical); 1x1: chunks in sequence.
    int tap = 0, cb = 0, kh_i = 0, kw_i = 0;
    if constexpr (SK) {
        if (kt0 > 0) {
            const int cb0 = a.ntaps > 1 ? kt0 / a.ntaps : kt0;
            const int tap0 = a.ntaps > 1 ? kt0 - cb0 * a.ntaps : 0;
            const int kh0 = tap0 / a.KW;
            cb = __builtin_amdgcn_readfirstlane(cb0);
            tap = __builtin_amdgcn_readfirstlane(tap0);
            kh_i = __builtin_amdgcn_readfirstlane(kh0);
            kw_i = __builtin_amdgcn_readfirstlane(tap0 - kh0 * a.KW);
        }
    }
    const int tapw = a.dil * a.ldx * 4, taph = a.dil * a.W * a.ldx * 4;     // byte displacement of one tap step

#define PEMP_DMA2(buf_)                                                                                           \
    do {                                                                                                          \
        v4f* Ad_ = As + (buf_) * BM * 8 + wave * 64;                                                              \
        v4f* Bd_ = Bs + (buf_) * BN * 8 + wave * 64;                                                              \
        const int sa_ = kh_i * taph + kw_i * tapw + cb * 128;                                                     \
        const int sb_ = (tap * a.Cin + cb * 32) * 4;                                                              \
        const int sh_ = 31 - tap;                                                                                 \
        const unsigned oob_ = PADV ? padv_off - (unsigned)(kh_i * taph + kw_i * tapw) : 0x80000000u;              \
        _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                                          \
            const unsigned vo_ = ((int)(a_inv[i] << sh_) < 0) ? oob_ : a_voff[i];                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(Ad_ + i * RPI * 8), 16, vo_, sa_, 0, 0);        \
        }                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < BL; ++i)                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(Bd_ + i * RPI * 8), 16, b_voff[i], sb_, 0, 0);  \
    } while (0)

    // branch-free and pinned to the scalar unit (inline asm: hipcc otherwise turns the selects into VALU code, the
    // SGPR offsets of the DMA instructions into VGPRs and every DMA into a readfirstlane loop); the steady-state
    // K step must stay one basic block
    // (the three loop-invariant inputs are produced by an asm statement with an "=s" result: hipcc has been seen to keep
    // a uniform value -- even the result of __builtin_amdgcn_readfirstlane -- in a VGPR and to print that VGPR into an
    // "s" operand)
    // HAZARDS the statement must cover itself (gfx940 / gfx950 wait-state rules that hipcc's hazard recogniser applies to the
    // instructions it schedules, NOT to the text of an inline-asm statement): "VALU writes a VGPR -> v_readlane /
    // v_readfirstlane reads it: 1 wait state" on the way in -- the "v" inputs are materialised by the compiler (v_mov /
    // v_cndmask from SGPRs) and may be the very instruction in front of the statement -- and "VALU writes an SGPR -> VALU reads it:
    // 2, v_readlane lane select: 4, VMEM reads it: 5 wait states" on the way out.  Hence the leading s_nop 0 and the trailing
    // s_nop 4.  Round 4's "tile 31" wrong results were exactly the first one: in ONE instantiation (128 x 128, 4 waves, padding
    // value, split-K) the scheduler put `v_cndmask_b32 v3, 0, 1, s[8:9]` (ntaps > 1 ? 1 : 0) directly in front of
    // `v_readfirstlane_b32 s8, v3`, the read returned v3's previous content, `multi` came out 0 and the K loop walked a 3 x 3
    // conv as if it were 1 x 1 (found round 5 by tracing the scalar state of the failing binary and inserting single s_nops
    // into its assembly: DESIGN.md section 4).
    int multi, s_kw, s_ntaps;
    asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %3\n\tv_readfirstlane_b32 %1, %4\n\tv_readfirstlane_b32 %2, %5\n\ts_nop 4"
                 : "=s"(multi), "=s"(s_kw), "=s"(s_ntaps)
                 : "v"(a.ntaps > 1 ? 1 : 0), "v"(a.KW), "v"(a.ntaps));
#define PEMP_ADVANCE2()                                                                                           \
    do {                                                                                                          \
        int wt_;                                                                                                  \
        asm volatile(                                                                                             \
            "s_add_u32 %0, %0, %5\n\t"      /* tap += multi                      */                                \
            "s_add_u32 %2, %2, %5\n\t"      /* kw  += multi                      */                                \
            "s_cmp_eq_u32 %2, %6\n\t"       /* kw == KW ?                        */                                \
            "s_cselect_b32 %2, 0, %2\n\t"   /*   kw = 0                          */                                \
            "s_addc_u32 %1, %1, 0\n\t"      /*   kh += 1                         */                                \
            "s_xor_b32 %4, %5, 1\n\t"       /* 1x1: wrap every step              */                                \
            "s_cmp_eq_u32 %0, %7\n\t"       /* tap == ntaps ?                    */                                \
            "s_cselect_b32 %4, 1, %4\n\t"                                                                          \
            "s_cmp_lg_u32 %4, 0\n\t"                                                                               \
            "s_cselect_b32 %0, 0, %0\n\t"   /*   tap = kh = kw = 0, next channel chunk */                          \
            "s_cselect_b32 %1, 0, %1\n\t"                                                                          \
            "s_cselect_b32 %2, 0, %2\n\t"                                                                          \
            "s_addc_u32 %3, %3, 0"                                                                                \
            : "+s"(tap), "+s"(kh_i), "+s"(kw_i), "+s"(cb), "=&s"(wt_)                                             \
            : "s"(multi), "s"(s_kw), "s"(s_ntaps)                                                    \
            : "scc");                                                                                             \
    } while (0)

    f32x16 acc[R16 ? 1 : TM][R16 ? 1 : TN];
#pragma unroll
    for (int mi = 0; mi < (R16 ? 1 : TM); ++mi)
#pragma unroll
        for (int ni = 0; ni < (R16 ? 1 : TN); ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    v4f acc16[R16 ? TN : 1];             // R16: one 16 x 16 accumulator per column tile (rows 4 (lane >> 4) + e, column lane & 15)
#pragma unroll
    for (int ni = 0; ni < (R16 ? TN : 1); ++ni) acc16[ni] = v4f{0.f, 0.f, 0.f, 0.f};

    PEMP_DMA2(0);
    if (nkl > 1) {
        PEMP_ADVANCE2();
        PEMP_DMA2(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");         // step 0 has landed, step 1 may still fly
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // fragment read positions: quad Q of row (.. + lr) sits at position Q ^ ((lr>>1)&7)
    const int rsw = (lr >> 1) & 7;
    const int arow = (wm0 + lr) * 8, brow = (wn0 + lr) * 8;
    v4f af[2][R16 ? 1 : TM], bf[2][R16 ? 1 : TN];
    // R16: lane (r16, g) feeds k slot g of v_mfma_f32_16x16x4_f32.  The two MFMAs of quarter j must see k = 8j + {0, 4, 1, 5} and
    // 8j + {2, 6, 3, 7} in their slots 0..3 (the chain order of the 32-row kernels): slot g reads quad 2j + (g & 1) of its row and
    // uses element (g >> 1) for the first MFMA, element 2 + (g >> 1) for the second -- one ds_read2_b32 (dwords +0, +2) per row.
    const int r16 = lane & 15, g16 = lane >> 4;
    const int rsw16 = (r16 >> 1) & 7;                 // wm0 / wn0 / 16 ni are multiples of 16: the row's swizzle is that of r16
    const int arow16 = (wm0 + r16) * 32 + (g16 >> 1), brow16 = (wn0 + r16) * 32 + (g16 >> 1);     // in floats
    float a16[2][2], b16[2][R16 ? TN : 1][2];

#define PEMP_READ(dst_, buf_, j_)                                                                                 \
    do {                                                                                                          \
        const v4f* Ab_ = As + (buf_) * BM * 8;                                                                    \
        const v4f* Bb_ = Bs + (buf_) * BN * 8;                                                                    \
        if constexpr (R16) {                                                                                      \
            const int pos_ = ((2 * (j_) + (g16 & 1)) ^ rsw16) * 4;                                                \
            const float* pa_ = (const float*)Ab_ + arow16 + pos_;                                                 \
            a16[dst_][0] = pa_[0];                                                                                \
            a16[dst_][1] = pa_[2];                                                                                \
            _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) {                                                   \
                const float* pb_ = (const float*)Bb_ + brow16 + ni * 512 + pos_;                                  \
                b16[dst_][ni][0] = pb_[0];                                                                        \
                b16[dst_][ni][1] = pb_[2];                                                                        \
            }                                                                                                     \
        } else {                                                                                                  \
            const int pos_ = (2 * (j_) + lh) ^ rsw;                                                               \
            _Pragma("unroll") for (int mi = 0; mi < TM; ++mi) af[dst_][mi] = Ab_[arow + mi * 256 + pos_];         \
            _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) bf[dst_][ni] = Bb_[brow + ni * 256 + pos_];         \
        }                                                                                                         \
    } while (0)

#define PEMP_MMA(src_)                                                                                            \
    do {                                                                                                          \
        if constexpr (R16) {      /* column tiles interleaved: no MFMA waits for the one in front of it */         \
            _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)    \
                acc16[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16[src_][h_], b16[src_][ni][h_], acc16[ni], 0, 0, 0); \
        } else                                                                                                    \
        _Pragma("unroll") for (int mi = 0; mi < TM; ++mi) _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) {     \
            const v4f av = af[src_][mi], bv = bf[src_][ni];                                                       \
            if constexpr (BF16) {                                                                                 \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),              \
                                                                      __builtin_bit_cast(bf16x8, bv), acc[mi][ni], 0, 0, 0); \
                continue;                                                                                         \
            }                                                                                                     \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[mi][ni], 0, 0, 0);                 \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[mi][ni], 0, 0, 0);                 \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[mi][ni], 0, 0, 0);                 \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[mi][ni], 0, 0, 0);                 \
        }                                                                                                         \
    } while (0)

    // One K step.  DMA_: issue the LDS-DMA of step kt+2 (into the buffer this step frees); NEXT_: read the first
    // fragments of step kt+1.  The steady-state body has both and no branch, so that the scheduler can place the reads
    // and the DMA issue between the MFMAs of the last quarter; the last two steps are peeled.
#define PEMP_STEP(buf_, DMA_, NEXT_)                                                                              \
    do {                                                                                                          \
        PEMP_READ(1, buf_, 1);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_MMA(0);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_READ(0, buf_, 2);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_MMA(1);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_READ(1, buf_, 3);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_MMA(0);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        /* every LDS read of `buf` by this wave has returned; this wave's DMA pieces of step kt+1 have landed */  \
        __builtin_amdgcn_s_waitcnt(0x0070);          /* vmcnt(0) lgkmcnt(0), visible to hipcc's own wait bookkeeping */ \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* (the LDS-DMA loads are not in that bookkeeping) */    \
        __builtin_amdgcn_s_barrier();   /* ... everybody's: `buf` is free for step kt+2, `buf^1` holds step kt+1 */ \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (NEXT_) PEMP_READ(0, (buf_) ^ 1, 0);                                                                   \
        if (DMA_) {                                                                                               \
            PEMP_ADVANCE2();                                                                                      \
            PEMP_DMA2(buf_);                                                                                      \
        }                                                                                                         \
        PEMP_MMA(1);                    /* the last quarter of step kt covers the reads / DMA issue above */      \
        if (DMA_) {                     /* ... placed BETWEEN its MFMAs: one LDS read / one DMA per MFMA slot */    \
            _Pragma("unroll") for (int k_ = 0; k_ < NMF; ++k_) {                                                  \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
                _Pragma("unroll") for (int q_ = 0; q_ < PER; ++q_) {                                              \
                    const int it_ = k_ * PER + q_;                                                                \
                    if (it_ < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
                    else if (it_ < NDS + NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                 \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    int kt = 0;
    PEMP_READ(0, 0, 0);
    for (; kt + 2 < nkl; ++kt) {
        const int buf = kt & 1;
        PEMP_STEP(buf, true, true);
    }
    if (kt + 1 < nkl) {
        const int buf = kt & 1;
        PEMP_STEP(buf, false, true);
        ++kt;
    }
    // Residual (shortcut) quads of small tiles are requested HERE, in front of the last K step's MFMAs, instead of inside
    // the epilogue where every tile would wait out a full memory latency between its LDS transpose and its stores.
    constexpr bool PRE = !R16 && TM * TN <= 2;
    v4f rpre[PRE ? TM * TN * 4 : 1];
#pragma unroll
    for (int i = 0; i < (PRE ? TM * TN * 4 : 1); ++i) rpre[i] = v4f{0.f, 0.f, 0.f, 0.f};
    if constexpr (PRE) {
        if (a.res) {
            const int rr_ = lane >> 3, c4_ = (lane & 7) * 4;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = m0 + wm0 + mi * 32 + rr_ + 8 * i, n = n0 + wn0 + ni * 32 + c4_;
                        rpre[(mi * TN + ni) * 4 + i] = m < a.M ? load_quad(a.res, (size_t)m * a.ldr + n, a.flags & PEMP_CONV_BF16_IO) : v4f{0.f, 0.f, 0.f, 0.f};
                    }
        }
    }
    {
        const int buf = kt & 1;
        PEMP_STEP(buf, false, false);
    }
#undef PEMP_STEP
#undef PEMP_DMA2
#undef PEMP_ADVANCE2
#undef PEMP_READ
#undef PEMP_MMA

    if constexpr (SK) {
        if (sk_r >= 0) {
            // partial accumulators in register order: float4 q of tile (mi, ni) of every lane, 1 KB per wave-instruction
            v4f* part = (v4f*)a.sk_ws + ((size_t)(sk_r * a.sk_S + piece) * NW + wave) * (TM * TN * 4 * 64) + lane;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(part + ((mi * TN + ni) * 4 + q) * 64), "v"(v) : "memory");
                    }
            // The hand-off.  Producer side: EVERY store of the handed-off bytes is a device-scope write-through store (`sc0 sc1`,
            // 16 B), every storing wave drains them (`s_waitcnt vmcnt(0)`), the workgroup's barrier, then ONE lane's returning
            // agent-scope atomic add on the tile's counter.  Consumer side (the workgroup whose add came last, told by the returned
            // value): that lane runs an AGENT-SCOPE ACQUIRE (`buffer_inv sc1`: this CU's vector L1) and waits for it before it
            // publishes the verdict through LDS, the other waves join it at the workgroup barrier, and every load of the bytes is
            // a `global_load_dwordx4 sc0 sc1` to registers.  That is MI355X_MICROARCH.md's "Consumer, always" form (one returned
            // atomic -> one agent acquire -> vmcnt(0) -> workgroup barrier -> loads) with the sc1-store producer form; the loads stay
            // sc1 on top of it.  Round 4 shipped this WITHOUT the acquire, claiming the guide's table of hand-offs measured with
            // sc1 loads in its place -- but that table's row is for hipMalloc memory at one workgroup per CU, and this workspace is
            // hipExtMallocWithFlags(uncached) with two workgroups per CU for the 128-row tiles: outside the row in two cells, so
            // the acquire stays (the guide's condition (4)).  It costs one L1 invalidate per SPLIT TILE in ONE workgroup --
            // measured (A/B of PEMP_SK_ACQUIRE, profiles/r05_splitk_acquire_ab.txt): +0.4 ... 1.1 us on the 35-70 us launches of a one-episode step, nothing at the training shapes --
            // not the release/acquire fence PAIR in every workgroup that round 2 measured at 118 -> 94 TFLOP/s.  (Rounds 2-3 used
            // plain stores and non-temporal loads on uncached memory, which held in every test until one launch of one run returned
            // a different tile: `nt` is not a coherent access.)  The workspace stays uncached device memory on top of that.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's partial stores have been acknowledged
            __syncthreads();                                               // ... everybody's
            int* flag = (int*)smem;
            if (tid == 0) {
                const int v = __hip_atomic_fetch_add(a.sk_cnt + sk_r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // only now this block counts as arrived
#if PEMP_SK_ACQUIRE
                if (v == a.sk_S - 1) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the invalidate has completed before anybody is released
                }
#endif
                *flag = v;
            }
            __syncthreads();
            const int arrived = *flag;
            __syncthreads();
            if (arrived != a.sk_S - 1) return;        // not the last piece of this tile: done
            if (tid == 0) __hip_atomic_store(a.sk_cnt + sk_r, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
            for (int pc = 0; pc < a.sk_S; ++pc) {     // pieces in ascending order, whichever arrived last: deterministic
                const v4f* src = (const v4f*)a.sk_ws + ((size_t)(sk_r * a.sk_S + pc) * NW + wave) * (TM * TN * 4 * 64) + lane;
                // two 32 x 32 sub-tiles (8 coherent 16-byte loads) in flight per wait; a lane's quads of one sub-tile lie 1 KB apart
                constexpr int NST = TM * TN;
#pragma unroll
                for (int st = 0; st < NST; st += 2) {
                    v4f t[8];
                    const v4f* q0 = src + (st * 4) * 64;
                    if constexpr (NST >= 2) {
                        const v4f* q1 = q0 + 4 * 64;
                        asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
                                     "global_load_dwordx4 %1, %8, off offset:1024 sc0 sc1\n\t"
                                     "global_load_dwordx4 %2, %8, off offset:2048 sc0 sc1\n\t"
                                     "global_load_dwordx4 %3, %8, off offset:3072 sc0 sc1\n\t"
                                     "global_load_dwordx4 %4, %9, off sc0 sc1\n\t"
                                     "global_load_dwordx4 %5, %9, off offset:1024 sc0 sc1\n\t"
                                     "global_load_dwordx4 %6, %9, off offset:2048 sc0 sc1\n\t"
                                     "global_load_dwordx4 %7, %9, off offset:3072 sc0 sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                                     : "v"(q0), "v"(q1)
                                     : "memory");
                    } else {
                        asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
                                     "global_load_dwordx4 %1, %4, off offset:1024 sc0 sc1\n\t"
                                     "global_load_dwordx4 %2, %4, off offset:2048 sc0 sc1\n\t"
                                     "global_load_dwordx4 %3, %4, off offset:3072 sc0 sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                                     : "v"(q0)
                                     : "memory");
                    }
#pragma unroll
                    for (int u = 0; u < (NST >= 2 ? 2 : 1); ++u) {
                        const int mi = (st + u) / TN, ni = (st + u) % TN;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[mi][ni][4 * q] += t[4 * u + q].x;
                            acc[mi][ni][4 * q + 1] += t[4 * u + q].y;
                            acc[mi][ni][4 * q + 2] += t[4 * u + q].z;
                            acc[mi][ni][4 * q + 3] += t[4 * u + q].w;
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: transpose through LDS.  No LDS read and no DMA is outstanding after the loop's last barrier. ----
    float* Rall = (float*)smem + NW * 1024;          // EPI: [NW][TN][2][8][32] sums of the waves, behind their transpose patches
    static_assert(EPI == 0 || NW * 1024 + NW * TN * 512 <= 64 * (BM + BN), "LDS: statistics area");
    static_assert(NW * 1024 <= 64 * (BM + BN), "LDS: one 4 KB transpose patch per wave");
    if constexpr (R16) conv_epilogue_r16<TN>(a, acc16, (float*)smem + wave * 1024, m0 + wm0, n0 + wn0, lane);
    else if constexpr (PRE) conv_epilogue_lds_pre<TM, TN, TM * TN * 4, EPI, DB>(a, acc, (float*)smem + wave * 1024, m0 + wm0, n0 + wn0, lane, rpre, Rall + wave * TN * 512);
    else if constexpr (EPI != 0) {
        const v4f none[1] = {{0.f, 0.f, 0.f, 0.f}};
        conv_epilogue_lds_pre<TM, TN, 1, EPI>(a, acc, (float*)smem + wave * 1024, m0 + wm0, n0 + wn0, lane, none, Rall + wave * TN * 512);
    } else conv_epilogue_lds<TM, TN, DB>(a, acc, (float*)smem + wave * 1024, m0 + wm0, n0 + wn0, lane);
    if constexpr (EPI != 0) {
        __syncthreads();
        conv_stats_store<BN, WGM, NW, TN>(a, Rall, bm, n0, tid);
    }
#endif
}

template <int BM, int BN, int WGM, int NW, bool PADV, int EPI = 0, bool SK = false, bool BF16 = false, bool DB = false, bool R16 = false>
__global__ __launch_bounds__(NW * 64) void conv_dma2_kernel(ConvArgs a) {
    conv_dma2_body<BM, BN, WGM, NW, PADV, EPI, SK, BF16, DB, R16>(a, blockIdx.x, gridDim.x);
}

// Several INDEPENDENT convs of the same tile shape in ONE launch: member i owns the blocks [first[i], first[i] + nblk[i]) (the
// first[] are multiples of 8, so that a block's XCD is the same function of its index inside the member as in a launch of its
// own; the few blocks in between return at once).  A one-episode step has 5202 feature rows -- 41 to 82 tiles per conv on 256
// CUs -- and convs that do not depend on each other (the dilated ASPP branches; a stage's downsample conv beside its conv1):
// together they fill the chip without splitting K and without one launch + drain per member.  Same tiles, same K order: every
// member's result is bit-identical to its own launch.
template <int BM, int BN, int WGM, int NW, bool PADV, bool R16 = false>
__global__ __launch_bounds__(NW * 64) void conv_dma2_group_kernel(ConvGroupArgs g) {
    int which = 0;
#pragma unroll
    for (int i = 1; i < CONV_GROUP_MAX; ++i) which += (i < g.n && (int)blockIdx.x >= g.first[i]) ? 1 : 0;
    const int bid = (int)blockIdx.x - g.first[which];
    if (bid >= g.nblk[which]) return;
    conv_dma2_body<BM, BN, WGM, NW, PADV, 0, false, false, false, R16>(g.a[which], bid, g.nblk[which]);
}

template <int BM, int BN, int WGM, int NW, bool R16 = false>
static int launch_dma2_group(ConvGroupArgs& g, bool padv, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(v4f);
    auto kern = padv ? conv_dma2_group_kernel<BM, BN, WGM, NW, true, R16> : conv_dma2_group_kernel<BM, BN, WGM, NW, false, R16>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    int grid = 0;
    for (int i = 0; i < g.n; ++i) {
        g.first[i] = grid;
        g.nblk[i] = cdiv(g.a[i].M, BM) * (g.a[i].Cout / BN);
        grid += (g.nblk[i] + 7) & ~7;
    }
    g.first[g.n] = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, g);
    return launch_status("conv_dma2/group");
}

// Hybrid launch (tile id 29) for convs of a FEW rounds: 32 x 32 wave tiles are the efficient shape, but T of them on 1024 SIMDs
// take ceil(T / 1024) rounds (a one-episode 256-channel conv: 1304 tiles, two rounds, the second 27 % full).  Here the rows that
// fill whole rounds go to the 64 x 64 tile and the remaining rows to 16-row wave tiles (half the work per K step) in the SAME
// grid: the SIMDs that would have taken a second 32 x 32 tile take a 16 x 32 one beside their first, 1.5 instead of 2 units per
// K step.  Both members run the K loop in the same order: bit-identical to every other variant.
static int g_simds_of[64] = {0};     // SIMD count per device id (filled on first use; racing fills write the same value)

template <bool PADV>
__global__ __launch_bounds__(256) void conv_dma2_hybrid_kernel(ConvGroupArgs g) {
    const int bid = (int)blockIdx.x;
    if (bid < g.first[1]) {
        if (bid < g.nblk[0]) conv_dma2_body<64, 64, 2, 4, PADV, 0, false, false, false, false>(g.a[0], bid, g.nblk[0]);
    } else {
        conv_dma2_body<32, 64, 2, 4, PADV, 0, false, false, false, true>(g.a[1], bid - g.first[1], g.nblk[1]);
    }
}

// rows of an M x Cout conv that go to the 64 x 64 tile in the hybrid launch (the rest: 16-row tiles); 0 = no such split
int conv_dma2_hybrid_rows(int M, int Cout) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;        // no device (CPU-side query): the MI355X count
    if (!g_simds_of[dev]) {
        int cus = 0;
        if (dev == 63 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        g_simds_of[dev] = 4 * cus;
    }
    const int g_simds = g_simds_of[dev];
    if (M <= 0 || Cout < 64 || Cout % 64) return 0;
    const int per32 = Cout / 32;                                 // 32 x 32 wave tiles per 32 output rows
    const int rounds = (int)(((long long)M / 32 * per32) / g_simds);        // whole rounds the 32-row tiles fill
    if (rounds < 1 || rounds > 8) return 0;                      // (many rounds: the quantisation loss is small anyway)
    const int rows_a = ((long long)rounds * g_simds / per32) * 32 / 64 * 64;
    const int rest = M - rows_a;
    if (rows_a <= 0 || rest <= 0 || (long long)cdiv(rest, 16) * per32 > g_simds) return 0;
    return rows_a;
}

int launch_conv_dma2_hybrid(const ConvArgs& a, hipStream_t st) {
    if (a.stats || (a.flags & PEMP_CONV_BF16_IO)) return -2;
    const int rows_a = conv_dma2_hybrid_rows(a.M, a.Cout);
    if (!rows_a) return -2;
    const int rest = a.M - rows_a;
    ConvGroupArgs g;
    g.n = 2;
    g.a[0] = a;
    g.a[0].M = rows_a;
    g.a[1] = a;
    g.a[1].bm_first = rows_a / 32;
    for (int i = 2; i < CONV_GROUP_MAX; ++i) g.a[i] = a;
    g.nblk[0] = (rows_a / 64) * (a.Cout / 64);
    g.nblk[1] = cdiv(rest, 32) * (a.Cout / 64);
    g.first[0] = 0;
    g.first[1] = (g.nblk[0] + 7) & ~7;
    g.first[2] = g.first[1] + g.nblk[1];
    const size_t lds = (size_t)2 * 8 * (64 + 64) * sizeof(v4f);
    auto kern = a.padv ? conv_dma2_hybrid_kernel<true> : conv_dma2_hybrid_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(g.first[2]), dim3(256), lds, st, g);
    return launch_status("conv_dma2/hybrid");
}

int launch_conv_dma2_group(int tile, ConvGroupArgs& g, hipStream_t st) {
    const bool padv = g.a[0].padv != nullptr;
    if (tile == 8) return launch_dma2_group<32, 64, 2, 4, true>(g, padv, st);
    if (tile == 7) return launch_dma2_group<256, 256, 4, 8>(g, padv, st);
    if (tile == 6) return launch_dma2_group<256, 128, 4, 8>(g, padv, st);
    if (tile == 4) return launch_dma2_group<128, 128, 4, 8>(g, padv, st);
    if (tile == 5) return launch_dma2_group<128, 64, 4, 8>(g, padv, st);
    if (tile == 1) return launch_dma2_group<128, 128, 2, 4>(g, padv, st);
    if (tile == 2) return launch_dma2_group<128, 64, 2, 4>(g, padv, st);
    return launch_dma2_group<64, 64, 2, 4>(g, padv, st);
}

template <int BM, int BN, int WGM, int NW>
static int launch_dma2(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(v4f);
    auto kern = a.stats ? (a.bz ? conv_dma2_kernel<BM, BN, WGM, NW, false, 2> : conv_dma2_kernel<BM, BN, WGM, NW, false, 1>)
                        : a.padv ? conv_dma2_kernel<BM, BN, WGM, NW, true> : conv_dma2_kernel<BM, BN, WGM, NW, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    const int grid = cdiv(a.M, BM) * (a.Cout / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    return launch_status("conv_dma2");
}

template <int BM, int BN, int WGM, int NW>
static int launch_dma2_bf16(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(v4f);
    auto kern = a.padv ? conv_dma2_kernel<BM, BN, WGM, NW, true, 0, false, true> : conv_dma2_kernel<BM, BN, WGM, NW, false, 0, false, true>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    const int grid = cdiv(a.M, BM) * (a.Cout / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    return launch_status("conv_dma2/bf16");
}

int launch_conv_dma2_bf16(int tile, const ConvArgs& a, hipStream_t st) {
    if (tile == 7) return launch_dma2_bf16<256, 256, 4, 8>(a, st);
    if (tile == 6) return launch_dma2_bf16<256, 128, 4, 8>(a, st);
    if (tile == 4) return launch_dma2_bf16<128, 128, 4, 8>(a, st);
    if (tile == 5) return launch_dma2_bf16<128, 64, 4, 8>(a, st);
    if (tile == 1) return launch_dma2_bf16<128, 128, 2, 4>(a, st);
    if (tile == 2) return launch_dma2_bf16<128, 64, 2, 4>(a, st);
    return launch_dma2_bf16<64, 64, 2, 4>(a, st);
}

// conv + DropBlock2D's row scaling in the epilogue (pemp_conv2d_dropblock_nhwc_f32): DB instantiations, unsplit and split-K
template <int BM, int BN, int WGM, int NW, bool SK>
static int launch_dma2_db(const ConvArgs& a, int grid, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(v4f);
    auto kern = conv_dma2_kernel<BM, BN, WGM, NW, false, 0, SK, false, true>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    return launch_status("conv_dma2/dropblock");
}

template <bool SK>
static int launch_db_tile(int tile, const ConvArgs& a, int grid, hipStream_t st) {
    if (tile == 7) return launch_dma2_db<256, 256, 4, 8, SK>(a, grid, st);
    if (tile == 6) return launch_dma2_db<256, 128, 4, 8, SK>(a, grid, st);
    if (tile == 4) return launch_dma2_db<128, 128, 4, 8, SK>(a, grid, st);
    if (tile == 5) return launch_dma2_db<128, 64, 4, 8, SK>(a, grid, st);
    if (tile == 1) return launch_dma2_db<128, 128, 2, 4, SK>(a, grid, st);
    if (tile == 2) return launch_dma2_db<128, 64, 2, 4, SK>(a, grid, st);
    return launch_dma2_db<64, 64, 2, 4, SK>(a, grid, st);
}

static void tile_shape(int tile, int& bm, int& bn);

int launch_conv_dma2_db(int tile, ConvArgs a, void* ws, size_t ws_bytes, bool split, hipStream_t st) {
    int bm, bn;
    tile_shape(tile, bm, bn);
    if (split) {
        const SplitKPlan p = conv_dma2_splitk_plan(tile, a);
        if (p.pieces >= 2) {
            if (!ws || ws_bytes < p.ws_bytes || ((uintptr_t)ws & 15)) {
                set_error("conv split-K: workspace of %zu bytes needed (16-byte aligned), got %zu", p.ws_bytes, ws_bytes);
                return -1;
            }
            a.sk_cnt = (int*)ws;
            a.sk_ws = (float*)((char*)ws + 1024);
            a.sk_full = p.full;
            a.sk_S = p.pieces;
            return launch_db_tile<true>(tile, a, p.full + p.split * p.pieces, st);
        }
    }
    return launch_db_tile<false>(tile, a, cdiv(a.M, bm) * (a.Cout / bn), st);
}

int conv_dma2_tile_rows(int tile) {        // BM of tile variant 1..8
    static const int bm[9] = {0, 128, 128, 64, 128, 128, 256, 256, 32};
    return tile >= 1 && tile <= 8 ? bm[tile] : 0;
}

static void tile_shape(int tile, int& bm, int& bn) {
    static const int shapes[9][2] = {{0, 0}, {128, 128}, {128, 64}, {64, 64}, {128, 128}, {128, 64}, {256, 128}, {256, 256}, {32, 64}};
    bm = shapes[tile][0];
    bn = shapes[tile][1];
}

// The tiles of a launch are dealt to the 256 CUs in rounds; T mod 256 tiles are left for a last, partly filled round (8 images
// of 51 x 51 pixels: 326 tiles of 128 x 128 for 256 output channels -- the chip is busy for two rounds and does the work
// of 1.27).  Those remainder tiles are split along K into as many pieces as keep the piece count <= 256, so that the last
// round is short instead of partly filled.
SplitKPlan conv_dma2_splitk_plan(int tile, const ConvArgs& a) {
    SplitKPlan p = {0, 0, 1, 0};
    if (tile < 1 || tile > 7 || tile == 3) return p;
    int bm, bn;
    tile_shape(tile, bm, bn);
    const int T = cdiv(a.M, bm) * (a.Cout / bn);
    const int rem = T % 256;
    p.full = T;
    if (rem == 0 || rem > 128) return p;
    int pieces = 256 / rem;
    if (pieces > a.nk / 4) pieces = a.nk / 4;
    if (pieces > 16) pieces = 16;
    if (pieces < 2) return p;
    p.full = T - rem;
    p.split = rem;
    p.pieces = pieces;
    p.ws_bytes = 1024 + (size_t)rem * pieces * bm * bn * sizeof(float);     // counters (<= 128 ints), then the partial tiles
    return p;
}

template <int BM, int BN, int WGM, int NW>
static int launch_dma2_sk(const ConvArgs& a, int grid, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(v4f);
    auto kern = a.stats ? (a.bz ? conv_dma2_kernel<BM, BN, WGM, NW, false, 2, true> : conv_dma2_kernel<BM, BN, WGM, NW, false, 1, true>)
                        : a.padv ? conv_dma2_kernel<BM, BN, WGM, NW, true, 0, true> : conv_dma2_kernel<BM, BN, WGM, NW, false, 0, true>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    return launch_status("conv_dma2/splitk");
}

int launch_conv_dma2_splitk(int tile, ConvArgs a, void* ws, size_t ws_bytes, hipStream_t st) {
    const SplitKPlan p = conv_dma2_splitk_plan(tile, a);
    if (p.pieces < 2 || (a.padv && a.stats)) return launch_conv_dma2(tile, a, st);       // nothing to split: the plain variant
    if (!ws || ws_bytes < p.ws_bytes || ((uintptr_t)ws & 15)) {
        set_error("conv split-K: workspace of %zu bytes needed (16-byte aligned), got %zu", p.ws_bytes, ws_bytes);
        return -1;
    }
    a.sk_cnt = (int*)ws;
    a.sk_ws = (float*)((char*)ws + 1024);
    a.sk_full = p.full;
    a.sk_S = p.pieces;
    const int grid = p.full + p.split * p.pieces;
    if (tile == 7) return launch_dma2_sk<256, 256, 4, 8>(a, grid, st);
    if (tile == 6) return launch_dma2_sk<256, 128, 4, 8>(a, grid, st);
    if (tile == 4) return launch_dma2_sk<128, 128, 4, 8>(a, grid, st);
    if (tile == 5) return launch_dma2_sk<128, 64, 4, 8>(a, grid, st);
    if (tile == 1) return launch_dma2_sk<128, 128, 2, 4>(a, grid, st);
    return launch_dma2_sk<128, 64, 2, 4>(a, grid, st);
}

// true when the geometry / operands fit this variant (the caller falls back to conv_dma.hip otherwise)
bool conv_dma2_supported(const ConvArgs& a) {
    if ((a.flags & PEMP_CONV_STEM4) || a.ntaps > 32 || (a.stats && a.padv)) return false;
    const long long xbytes = ((long long)a.N * a.H * a.W + (long long)a.pad * a.W + a.pad + (long long)a.dil * (a.KH - 1) * a.W +
                              (long long)a.dil * (a.KW - 1)) * a.ldx * 4;
    const long long wbytes = (long long)a.Cout * a.Kpad * 4;
    if (a.padv) {           // the padding vector must sit behind the activations, inside the 2 GiB window of their descriptor,
                            // and far enough in that subtracting the largest tap displacement leaves a non-negative offset
        const long long behind = (const char*)a.padv - (const char*)a.x - (long long)a.N * a.H * a.W * a.ldx * 4;
        const long long d = (const char*)a.padv - (const char*)a.x + ((long long)a.pad * a.W + a.pad) * a.ldx * 4;
        const long long tapmax = ((long long)a.dil * (a.KH - 1) * a.W + (long long)a.dil * (a.KW - 1)) * a.ldx * 4;
        if (behind < 0 || d < tapmax || d + (long long)a.Cin * 4 >= (1ll << 31)) return false;
    }
    return xbytes < (1ll << 31) && wbytes < (1ll << 31);
}

// tile 8: the 16-row variant (32 x 64 block, 16 x 32 wave tiles on v_mfma_f32_16x16x4_f32); plain epilogue only
static int launch_dma2_r16(const ConvArgs& a, hipStream_t st) {
    if (a.stats) {
        set_error("conv_dma2: the 16-row tile has no statistics epilogue");
        return -1;
    }
    auto kern = a.padv ? conv_dma2_kernel<32, 64, 2, 4, true, 0, false, false, false, true>
                       : conv_dma2_kernel<32, 64, 2, 4, false, 0, false, false, false, true>;
    const size_t lds = (size_t)2 * 8 * (32 + 64) * sizeof(v4f);
    const int grid = cdiv(a.M, 32) * (a.Cout / 64);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return launch_status("conv_dma2/r16");
}

int launch_conv_dma2(int tile, const ConvArgs& a, hipStream_t st) {
    if (tile == 8) return launch_dma2_r16(a, st);
    if (tile == 7) return launch_dma2<256, 256, 4, 8>(a, st);
    if (tile == 6) return launch_dma2<256, 128, 4, 8>(a, st);
    if (tile == 4) return launch_dma2<128, 128, 4, 8>(a, st);
    if (tile == 5) return launch_dma2<128, 64, 4, 8>(a, st);
    if (tile == 1) return launch_dma2<128, 128, 2, 4>(a, st);
    if (tile == 2) return launch_dma2<128, 64, 2, 4>(a, st);
    return launch_dma2<64, 64, 2, 4>(a, st);
}

}  // namespace pemp
