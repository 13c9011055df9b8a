// Weight gradient of the convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32:
//
//     dW[co][tap][ci] = sum_m  g[m][co] * x[pix(m, tap)][ci]          (m = output pixel)
//
//   GEMM view: rows = co, columns = (tap, ci) (the KRSC weight row), reduction = m.
//   Block = 256 threads = 2x2 waves, output tile 64 co x 64 columns (one tap, 64 input channels;
//   for the NHWC4 stem: 16 taps x 4 channels).  One reduction step = 32 pixels: the g tile
//   [32 px][64 co] and the gathered x tile [32 px][64 ci] go global -> LDS by LDS-DMA in their
//   natural (pixel-major) layout -- the MFMA contracts over pixels, so lane (r = l&31, k = l>>5)
//   reads ONE float g[px 2t+k][co r] / x[px 2t+k][ci r] per MFMA: 32 consecutive dwords per half
//   wave, conflict-free ds_read_b32, no transpose anywhere.
//
//   The reduction over M is split over blockIdx.z; partial tiles go to a workspace and a second
//   kernel adds them in a fixed order (deterministic, no atomics).
#include "conv_common.h"
#include <stdlib.h>
#include <algorithm>

namespace pemp {

static __device__ __attribute__((aligned(16))) float w_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct WgradArgs {
    const float* x;    // input activations  [N,H,W,ldx]
    const float* g;    // output gradient    [M][ldg]
    float* out;        // dW (nsplit == 1) or workspace [nsplit][Cout][Kpad]
    int N, H, W, Cin, ldx, Ho, Wo, Cout, ldg, KH, KW, stride, pad, dil, Kpad;
    int M, HoWo, ntaps, cin_tiles, steps_total, steps_per_split, nsplit, stem;
    int xcd, nsplit_grid;      // A/B switch (PEMP_WGRAD_XCD=0: tile-major block order, splits of a tile on consecutive ids)
    int gx, gy;        // tiles of the weight matrix along Cout / along its row (conv_wgrad2_kernel decodes a 1-D grid)
};

template <bool STEM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) v4f smem[];
    v4f* Gs = smem;                 // [2][32 px][16 quads]
    v4f* Xs = smem + 2 * 32 * 16;   // [2][32 px][16 quads]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;       // wave's 32x32 sub-tile: rows co, cols k

    const int co0 = blockIdx.x * 64;
    const int ky = blockIdx.y;                     // 64-column tile of the weight row
    const int split = blockIdx.z;
    const int s_begin = split * a.steps_per_split;
    const int s_end = min(s_begin + a.steps_per_split, a.steps_total);

    // loader role: thread fetches quad q of rows rowA and rowA + 4 (per tile)
    const int q = lane & 15;
    const int rloc = 8 * wave + (lane >> 4);       // + 4 i
    const float* zero = w_zero16;

    int tap, ci0;
    if (STEM) {
        tap = ky * 16 + q;                         // each quad is one tap (4 channels)
        ci0 = 0;
    } else {
        tap = ky / a.cin_tiles;
        ci0 = (ky - tap * a.cin_tiles) * 64;
    }
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int dh = kh * a.dil - a.pad, dw = kw * a.dil - a.pad;
    const bool tap_ok = tap < a.ntaps;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

#define PEMP_WG_DMA(step_, buf_)                                                                       \
    do {                                                                                               \
        v4f* Gd_ = Gs + (buf_) * 512 + wave * 128;                                                     \
        v4f* Xd_ = Xs + (buf_) * 512 + wave * 128;                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
            const int m = (step_) * 32 + rloc + 4 * i;                                                 \
            const bool mok = m < a.M;                                                                  \
            const float* gs = mok ? a.g + (size_t)m * a.ldg + co0 + q * 4 : zero;                      \
            __builtin_amdgcn_global_load_lds((gptr_t)gs, (lptr_t)(Gd_ + i * 64), 16, 0, 0);           \
            const int mm = mok ? m : 0;                                                                \
            const int img = mm / a.HoWo;                                                               \
            const int rem = mm - img * a.HoWo;                                                         \
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;                                           \
            const int hi = ho * a.stride + dh, wi = wo * a.stride + dw;                                \
            const bool ok = mok && tap_ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W; \
            const float* xs = ok ? a.x + ((size_t)(img * a.H + hi) * a.W + wi) * a.ldx + ci0 + (STEM ? 0 : q * 4) \
                                 : zero;                                                               \
            __builtin_amdgcn_global_load_lds((gptr_t)xs, (lptr_t)(Xd_ + i * 64), 16, 0, 0);           \
        }                                                                                              \
    } while (0)

    if (s_begin < s_end) {
        PEMP_WG_DMA(s_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const float* Gf = (const float*)Gs;
    const float* Xf = (const float*)Xs;
    const int gcol = wr * 32 + lr, xcol = wc * 32 + lr;
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) PEMP_WG_DMA(s + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const float* Gb = Gf + buf * 2048 + lh * 64 + gcol;
        const float* Xb = Xf + buf * 2048 + lh * 64 + xcol;
        float ga[2], xa[2];
        ga[0] = Gb[0];
        xa[0] = Xb[0];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < 15) {
                ga[(t + 1) & 1] = Gb[(t + 1) * 128];
                xa[(t + 1) & 1] = Xb[(t + 1) * 128];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[t & 1], xa[t & 1], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef PEMP_WG_DMA

    // D[i][j]: i = co (rows), j = lane&31 = column
    float* out = a.out + (size_t)split * a.Cout * a.Kpad;
    const int col = ky * 64 + wc * 32 + lr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = co0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (co < a.Cout && col < a.Kpad) out[(size_t)co * a.Kpad + col] = acc[e];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 128 co x 128 columns per block (Cout and Cin multiples of 128): 2x2 waves of 64 x 64 (four MFMA tiles each), so one
// reduction step of 32 pixels carries 64 MFMAs per wave between barriers (16 in the 64x64 kernel), every operand
// float read from LDS feeds two MFMAs, and the L2 -> LDS traffic per flop is halved.  Same pixel-major LDS image,
// same conflict-free ds_read_b32 pattern; the reduction over pixels runs in the same order within a split.
__global__ __launch_bounds__(256) void conv_wgrad128_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) v4f smem[];
    v4f* Gs = smem;                  // [2][32 px][32 quads]
    v4f* Xs = smem + 2 * 32 * 32;    // [2][32 px][32 quads]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;        // wave's 64 x 64 sub-tile: rows co, columns k

    const int co0 = blockIdx.x * 128;
    const int ky = blockIdx.y;                      // 128-column tile of the weight row: one tap, 128 input channels
    const int split = blockIdx.z;
    const int s_begin = split * a.steps_per_split;
    const int s_end = min(s_begin + a.steps_per_split, a.steps_total);

    // loader role: thread fetches quad q (of 32) of pixel rows rloc + 8 i, i = 0..3
    const int q = lane & 31;
    const int rloc = 2 * wave + (lane >> 5);
    const float* zero = w_zero16;

    const int cin_tiles = a.Cin / 128;
    const int tap = ky / cin_tiles;
    const int ci0 = (ky - tap * cin_tiles) * 128;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int dh = kh * a.dil - a.pad, dw = kw * a.dil - a.pad;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // one DMA wave-instruction moves 2 pixel rows x 512 B; wave w owns rows 2w, 2w+1 (+ 8 i)
#define PEMP_WG_DMA(step_, buf_)                                                                       \
    do {                                                                                               \
        v4f* Gd_ = Gs + (buf_) * 1024 + wave * 64;                                                     \
        v4f* Xd_ = Xs + (buf_) * 1024 + wave * 64;                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const int m = (step_) * 32 + rloc + 8 * i;                                                 \
            const bool mok = m < a.M;                                                                  \
            const float* gs = mok ? a.g + (size_t)m * a.ldg + co0 + q * 4 : zero;                      \
            __builtin_amdgcn_global_load_lds((gptr_t)gs, (lptr_t)(Gd_ + i * 256), 16, 0, 0);          \
            const int mm = mok ? m : 0;                                                                \
            const int img = mm / a.HoWo;                                                               \
            const int rem = mm - img * a.HoWo;                                                         \
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;                                           \
            const int hi = ho * a.stride + dh, wi = wo * a.stride + dw;                                \
            const bool ok = mok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;       \
            const float* xs = ok ? a.x + ((size_t)(img * a.H + hi) * a.W + wi) * a.ldx + ci0 + q * 4 : zero; \
            __builtin_amdgcn_global_load_lds((gptr_t)xs, (lptr_t)(Xd_ + i * 256), 16, 0, 0);          \
        }                                                                                              \
    } while (0)

    if (s_begin < s_end) {
        PEMP_WG_DMA(s_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const float* Gf = (const float*)Gs;
    const float* Xf = (const float*)Xs;
    const int gcol = wr * 64 + lr, xcol = wc * 64 + lr;
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) PEMP_WG_DMA(s + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const float* Gb = Gf + buf * 4096 + lh * 128 + gcol;     // pixel 2t + lh, row stride 128 floats
        const float* Xb = Xf + buf * 4096 + lh * 128 + xcol;
        float ga[2][2], xa[2][2];
        ga[0][0] = Gb[0]; ga[0][1] = Gb[32];
        xa[0][0] = Xb[0]; xa[0][1] = Xb[32];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < 15) {
                ga[(t + 1) & 1][0] = Gb[(t + 1) * 256];
                ga[(t + 1) & 1][1] = Gb[(t + 1) * 256 + 32];
                xa[(t + 1) & 1][0] = Xb[(t + 1) * 256];
                xa[(t + 1) & 1][1] = Xb[(t + 1) * 256 + 32];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[t & 1][i], xa[t & 1][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef PEMP_WG_DMA

    float* out = a.out + (size_t)split * a.Cout * a.Kpad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = ky * 128 + wc * 64 + j * 32 + lr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                out[(size_t)co * a.Kpad + col] = acc[i][j][e];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Second generation of the two kernels above (same GEMM view, LDS image, fragment reads and MFMA order: bit-identical
// partial tiles), rebuilt the way conv_dma2.hip rebuilt the forward kernel:
//   * both operands come in through `buffer_load_dwordx4 ... offen lds`.  The gradient rows are consecutive, so their
//     per-lane offset is a constant and the reduction step lives in the SGPR offset; the activation rows follow the
//     output pixel through (image, ho, wo) INCREMENTALLY (+32 pixels per step: one conditional wrap of wo, one of ho)
//     instead of two integer divisions per row and step; rows / taps outside the image or beyond M get offset 2^31 and
//     the buffer range check writes zeros;
//   * one barrier per step, placed before the last quarter of the step's MFMAs (operands already in registers): the
//     next step's first reads and the DMA of the step after that are issued in their shadow.
// TW = 64 / 128: tile TW co x TW weight columns per block of 2x2 waves.  Needs Wo >= 32 (one wrap per step) and
// operands below 2 GiB; the stem keeps conv_wgrad_kernel<true>.
template <int TW>
__global__ __launch_bounds__(256) void conv_wgrad2_kernel(WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int QPR = TW / 4;            // quads per pixel row of a tile
    constexpr int RPR = 256 / QPR;         // pixel rows per DMA round of the block
    constexpr int NR = 32 / RPR;           // DMA rounds per operand and step
    constexpr int T = TW / 64;             // 32x32 MFMA tiles per wave and dimension
    constexpr int WNDS = 8;                // LDS reads per quarter: 4 sub-steps x 2 operands (b64 for TW = 128, b32 for TW = 64)
    constexpr int WPER = (WNDS + 2 * NR + 4 * T * T - 1) / (4 * T * T);
    extern __shared__ __attribute__((aligned(16))) v4f smem[];
    v4f* Gs = smem;                        // [2][32 px][QPR]
    v4f* Xs = smem + 2 * 32 * QPR;         // [2][32 px][QPR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    // Block -> (split, tile), XCD-aware: the blocks of one split read the same pixel rows of g and x (every (co, k) tile of the
    // weight matrix needs them), so they should meet in ONE XCD's L2.  Consecutive block ids go round-robin over the 8 XCDs;
    // xcd_tile_order hands every XCD a contiguous range of the split-major logical order.  (Measured before: 10.4 GB of L2
    // fills per training step for 42 MB operands per launch -- each XCD fetched every row for itself.)
    const int ntile = a.gx * a.gy;
    const int li = a.xcd ? xcd_tile_order(blockIdx.x, gridDim.x) : blockIdx.x;
    const int split = a.xcd ? li / ntile : li % a.nsplit_grid;
    const int tl = a.xcd ? li - split * ntile : li / a.nsplit_grid;
    const int ky = tl / a.gx;              // TW-column tile of the weight row: one tap, TW input channels
    const int co0 = (tl - ky * a.gx) * TW;
    const int s_begin = split * a.steps_per_split;
    const int s_end = min(s_begin + a.steps_per_split, a.steps_total);
    const int nsteps = s_end - s_begin;

    const int cin_tiles = a.Cin / TW;
    const int tap = ky / cin_tiles;
    const int ci0 = (ky - tap * cin_tiles) * TW;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int dh = kh * a.dil - a.pad, dw = kw * a.dil - a.pad;

    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0x80000000u, 0x00020000);

    // loader role: thread (row = tid / QPR, q = tid % QPR) fetches quad q of pixel rows row + RPR i of the step
    const int q = tid % QPR, row = tid / QPR;
    unsigned g_voff[NR];
    int x_m[NR], x_hi[NR], x_wi[NR];       // output pixel index; input row / column of this tap for that pixel (may be outside)
    unsigned x_off[NR];                    // byte offset of (img, hi, wi) + channel quad; only used when (hi, wi) is inside
    const int ldx4 = a.ldx * 4;
    const int sW = a.Wo * a.stride, sH = a.Ho * a.stride;
    const int inc32 = 32 * a.stride * ldx4;                                 // 32 output pixels further in the same row
    const int incw = (a.stride * a.W - a.Wo * a.stride) * ldx4;             // wo wrapped: next output row
    const int inch = (a.H * a.W - a.Ho * a.stride * a.W) * ldx4;            // ho wrapped: next image
    const int wi_hi = dw + sW, hi_hi = dh + sH;                             // first wi / hi beyond the last output column / row
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int m = s_begin * 32 + row + RPR * i;
        g_voff[i] = (unsigned)((row + RPR * i) * a.ldg + co0 + q * 4) * 4u;
        const int mm = m < a.M ? m : 0;
        const int img = mm / a.HoWo;
        const int rem = mm - img * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        x_m[i] = m;
        x_hi[i] = ho * a.stride + dh;
        x_wi[i] = wo * a.stride + dw;
        // may be "negative" (wraps) for out-of-image taps; only used when the tap is inside the image
        x_off[i] = (unsigned)((((img * a.H + ho * a.stride + dh) * a.W + wo * a.stride + dw) * a.ldx + ci0 + q * 4) * 4);
    }
    // The gradient rows of a step are consecutive: their step dependence is ONE wave-uniform byte offset.  It must live in an
    // SGPR (inline asm: left to hipcc it sits in a VGPR and every LDS-DMA that uses it as soffset becomes a readfirstlane
    // loop -- four of them per step, which also keeps the scheduler from placing the DMA issue between the MFMAs).
    int s_g0, s_ginc;                      // soffset of step k of this block = s_g0 + k * s_ginc, computed on the scalar unit
    // (leading s_nop 0 / trailing s_nop 4: the gfx950 wait states "VALU writes VGPR -> v_readfirstlane reads it" and "VALU writes
    // SGPR -> VALU / VMEM reads it", which hipcc does not insert around the text of an asm statement -- see conv_dma2.hip)
    asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 4"
                 : "=s"(s_g0), "=s"(s_ginc)
                 : "v"(s_begin * 32 * a.ldg * 4), "v"(32 * a.ldg * 4));

#define PEMP_WG2_DMA(buf_, k_)                                                                                      \
    do {                                                                                                          \
        v4f* Gd_ = Gs + (buf_) * 32 * QPR + wave * 64;                                                            \
        v4f* Xd_ = Xs + (buf_) * 32 * QPR + wave * 64;                                                            \
        const int sg_ = s_g0 + (k_) * s_ginc;                                                                     \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) {                                                          \
            const bool mok = x_m[i] < a.M;                                                                        \
            const bool ok = mok & ((unsigned)x_hi[i] < (unsigned)a.H) & ((unsigned)x_wi[i] < (unsigned)a.W);      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lptr_t)(Gd_ + i * 256), 16, mok ? g_voff[i] : 0x80000000u, sg_, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(Xd_ + i * 256), 16, ok ? x_off[i] : 0x80000000u, 0, 0, 0);    \
            /* advance this row by 32 output pixels (Wo >= 32: at most one wrap of wo, then at most one of ho) */ \
            x_m[i] += 32;                                                                                         \
            int wi_ = x_wi[i] + 32 * a.stride;                                                                    \
            unsigned off_ = x_off[i] + (unsigned)inc32;                                                           \
            const bool c1 = wi_ >= wi_hi;                                                                         \
            wi_ = c1 ? wi_ - sW : wi_;                                                                            \
            off_ += c1 ? (unsigned)incw : 0u;                                                                     \
            int hi_ = x_hi[i] + (c1 ? a.stride : 0);                                                              \
            const bool c2 = hi_ >= hi_hi;                                                                         \
            hi_ = c2 ? hi_ - sH : hi_;                                                                            \
            off_ += c2 ? (unsigned)inch : 0u;                                                                     \
            x_wi[i] = wi_;                                                                                        \
            x_hi[i] = hi_;                                                                                        \
            x_off[i] = off_;                                                                                      \
        }                                                                                                         \
    } while (0)

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nsteps > 0) {
        PEMP_WG2_DMA(0, 0);
        if (nsteps > 1) {
            PEMP_WG2_DMA(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NR) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // Fragment reads.  A wave's T tiles along a dimension are INTERLEAVED channels (tile i = channels T r + i of the wave's
    // TW / 2, r = MFMA row): lane r then needs T adjacent floats of a pixel row -- one ds_read_b64 for the 128 x 128 tile,
    // with an immediate offset per (step quarter, sub-step): no per-read address arithmetic.  Which accumulator row holds
    // which channel only matters to the store at the end; every output element is the same sum in the same order.
    typedef float fragT __attribute__((ext_vector_type(T == 2 ? 2 : 1)));
    const int gcol = wr * (TW / 2) + T * lr, xcol = wc * (TW / 2) + T * lr;
    fragT ga[2][4], xa[2][4];              // [register buffer][t within the quarter]: the wave's T tiles

    // quarter Q of a step = reduction sub-steps t = 4Q .. 4Q+3 (pixel pairs 2t, 2t+1)
#define PEMP_WG2_READ(dst_, buf_, Q_)                                                                             \
    do {                                                                                                          \
        const float* Gb_ = (const float*)Gs + (buf_) * 32 * TW + lh * TW + gcol;                                  \
        const float* Xb_ = (const float*)Xs + (buf_) * 32 * TW + lh * TW + xcol;                                  \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                        \
            ga[dst_][t_] = *(const fragT*)(Gb_ + (4 * (Q_) + t_) * 2 * TW);                                       \
            xa[dst_][t_] = *(const fragT*)(Xb_ + (4 * (Q_) + t_) * 2 * TW);                                       \
        }                                                                                                         \
    } while (0)
#define PEMP_WG2_MMA(src_)                                                                                        \
    do {                                                                                                          \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) _Pragma("unroll") for (int i_ = 0; i_ < T; ++i_)         \
            _Pragma("unroll") for (int j_ = 0; j_ < T; ++j_)                                                      \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[src_][t_][i_], xa[src_][t_][j_], acc[i_][j_], 0, 0, 0); \
    } while (0)
#define PEMP_WG2_STEP(buf_, DMA_, NEXT_, k_)                                                                         \
    do {                                                                                                          \
        PEMP_WG2_READ(1, buf_, 1);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_WG2_MMA(0);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_WG2_READ(0, buf_, 2);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_WG2_MMA(1);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_WG2_READ(1, buf_, 3);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        PEMP_WG2_MMA(0);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        __builtin_amdgcn_s_waitcnt(0x0070);                /* vmcnt(0) lgkmcnt(0) */                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
        __builtin_amdgcn_s_barrier();                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (NEXT_) PEMP_WG2_READ(0, (buf_) ^ 1, 0);                                                               \
        if (DMA_) PEMP_WG2_DMA(buf_, (k_) + 2);                                                                          \
        PEMP_WG2_MMA(1);                                                                                          \
        if (DMA_) {                     /* one LDS read / one DMA between consecutive MFMAs of the last quarter */ \
            _Pragma("unroll") for (int k_ = 0; k_ < 4 * T * T; ++k_) {                                            \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
                _Pragma("unroll") for (int q_ = 0; q_ < WPER; ++q_) {                                             \
                    const int it_ = k_ * WPER + q_;                                                               \
                    if (it_ < WNDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
                    else if (it_ < WNDS + 2 * NR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)

    if (nsteps > 0) {
        PEMP_WG2_READ(0, 0, 0);
        int k = 0;
        for (; k + 2 < nsteps; ++k) {
            const int buf = k & 1;
            PEMP_WG2_STEP(buf, true, true, k);
        }
        if (k + 1 < nsteps) {
            const int buf = k & 1;
            PEMP_WG2_STEP(buf, false, true, k);
            ++k;
        }
        {
            const int buf = k & 1;
            PEMP_WG2_STEP(buf, false, false, k);
        }
    }
#undef PEMP_WG2_STEP
#undef PEMP_WG2_MMA
#undef PEMP_WG2_READ
#undef PEMP_WG2_DMA

    // accumulator tile (i, j), MFMA row r / column lane: output channel co0 + wr TW/2 + T r + i, weight column
    // ky TW + wc TW/2 + T lr + j (the interleaved assignment of the fragment reads): a lane's T column tiles are adjacent
    // floats of one weight row -- one 8-byte store for the 128 x 128 tile, 256 contiguous bytes per half wave
    float* out = a.out + (size_t)split * a.Cout * a.Kpad;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wr * (TW / 2) + T * ((e & 3) + 8 * (e >> 2) + 4 * lh) + i;
            float* dst = out + (size_t)co * a.Kpad + ky * TW + wc * (TW / 2) + T * lr;
            if constexpr (T == 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                *(f2*)dst = f2{acc[i][0][e], acc[i][1][e]};
            } else {
                dst[0] = acc[i][0][e];
            }
        }
#endif
}

// dw = sum over the K-splits of the partial weight gradients, split 0 first (fixed order).  Eight partials are loaded before
// they are added: the loop is latency-bound otherwise (one L2 / HBM round trip per split).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long n,
                                                           int nsplit, int accumulate) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n / 4; i += (long long)gridDim.x * blockDim.x) {
        float4 s = ((const float4*)ws)[i];
        int k = 1;
        for (; k + 8 <= nsplit; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ((const float4*)(ws + (size_t)(k + u) * n))[i];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
            }
        }
        for (; k < nsplit; ++k) {
            const float4 v = ((const float4*)(ws + (size_t)k * n))[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (accumulate) {
            const float4 o = ((float4*)dw)[i];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        ((float4*)dw)[i] = s;
    }
}

}  // namespace pemp

using namespace pemp;

static int pick_split(int tiles, int steps, int want = 0) {
    // aim for ~768 blocks (3 per CU) with at least 8 reduction steps each.  The kernel shares the chip with the
    // input-gradient chain on the other stream, so it need not fill it alone, and every split costs a [Cout][K] partial
    // that the reduce pass has to read again: 1536 / 1024 / 768 / 512 / 384 blocks -> 20.88 / 20.78 / 20.53 / 20.59 /
    // 20.76 ms per training step.
    static const int target = getenv("PEMP_WGRAD_BLOCKS") ? atoi(getenv("PEMP_WGRAD_BLOCKS")) : 768;      // tuning knob
    int s = (want > 0 ? want : target) / tiles;        // want: the caller's block count (desc.tile >> 8; the Python side times a
                                                       // few).  Rounded DOWN: two 64 KB blocks fit a CU, so 512 / 1024 blocks
                                                       // are whole rounds of the chip and a few blocks more would start another
    if (s > steps / 8) s = steps / 8;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return s;
}

// 128 x 128 tiles where they measured faster on MI355X at the training shapes (8 images, 51 x 51): the 3x3 layers with
// >= 256 output channels (84 -> 93 TFLOP/s) and the large 1x1 layers; the small-output 1x1 layers keep the 64 x 64
// kernel, whose 4x more blocks per weight matrix fill the chip better.
static bool wgrad_big_tiles(const pemp_conv_desc* d) {
    if ((d->flags & PEMP_CONV_STEM4) || d->Cin % 128 || d->Cout % 128) return false;
    if ((d->tile & 255) == 2) return true;          // the caller's choice (it timed both): 2 = 128 x 128, 3 = 64 x 64
    if ((d->tile & 255) == 3) return false;
    if (d->Cout < 256) return false;
    return d->KH * d->KW > 1 || (long long)d->Cin * d->Cout >= 512ll * 1024;
}

extern "C" size_t pemp_conv2d_wgrad_workspace_bytes(const pemp_conv_desc* d) {
    if (!d) return 0;
    const int M = d->N * d->Ho * d->Wo;
    const bool big = wgrad_big_tiles(d);
    const int tw = big ? 128 : 64;
    const int tiles = cdiv(d->Cout, tw) * (d->Kpad / tw > 0 ? cdiv(d->Kpad, tw) : 1);
    const int split = pick_split(tiles, cdiv(M, 32), d->tile >> 8);
    return (size_t)split * d->Cout * d->Kpad * sizeof(float) + 256;
}

extern "C" int pemp_conv2d_wgrad_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* g, float* dw,
                                          int accumulate, void* ws, size_t ws_bytes, void* stream) {
    PEMP_REQUIRE(d && x && g && dw, "wgrad: null pointer");
    const bool stem = d->flags & PEMP_CONV_STEM4;
    const int ntaps = d->KH * d->KW;
    PEMP_REQUIRE(d->Cout % 64 == 0, "wgrad: Cout=%d must be a multiple of 64", d->Cout);
    PEMP_REQUIRE(d->ldy >= d->Cout && d->ldy % 4 == 0, "wgrad: ldy (gradient stride) must be >= Cout and x4");
    PEMP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)dw & 15) == 0, "wgrad: pointers must be 16-byte aligned");
    if (stem) {
        PEMP_REQUIRE(d->Cin == 4 && d->ldx == 4 && d->Kpad % 64 == 0 && d->Kpad >= ntaps * 4, "wgrad: STEM4 needs NHWC4 input and Kpad %% 64 == 0");
    } else {
        PEMP_REQUIRE(d->Cin % 64 == 0 && d->ldx >= d->Cin && d->ldx % 4 == 0, "wgrad: Cin=%d must be a multiple of 64", d->Cin);
        PEMP_REQUIRE(d->Kpad == ntaps * d->Cin, "wgrad: Kpad must equal KH*KW*Cin");
    }
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    PEMP_REQUIRE(ho == d->Ho && wo == d->Wo, "wgrad: Ho/Wo do not match geometry");
    WgradArgs a;
    a.x = x; a.g = g;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.ldg = d->ldy; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.Kpad = d->Kpad;
    a.HoWo = d->Ho * d->Wo;
    a.M = d->N * a.HoWo;
    a.ntaps = ntaps;
    a.cin_tiles = stem ? 1 : d->Cin / 64;
    a.steps_total = cdiv(a.M, 32);
    a.stem = stem;
    const bool big = wgrad_big_tiles(d);
    const int tw = big ? 128 : 64;
    const int tiles_k = d->Kpad / tw;
    const int tiles = (d->Cout / tw) * tiles_k;
    a.nsplit = pick_split(tiles, a.steps_total, d->tile >> 8);
    a.steps_per_split = cdiv(a.steps_total, a.nsplit);
    a.nsplit = cdiv(a.steps_total, a.steps_per_split);
    const bool direct = a.nsplit == 1 && !accumulate;
    if (!direct) {
        PEMP_REQUIRE(ws && ws_bytes >= (size_t)a.nsplit * d->Cout * d->Kpad * sizeof(float), "wgrad: workspace too small");
        PEMP_REQUIRE(((uintptr_t)ws & 15) == 0, "wgrad: workspace must be 16-byte aligned");
    }
    a.out = direct ? dw : (float*)ws;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(d->Cout / tw, tiles_k, a.nsplit);
    // second-generation kernels: one wrap of wo per 32-pixel step, buffer offsets below 2 GiB (desc.tile = 1 asks for the
    // first generation: the parity tests compare the two bit for bit)
    const bool v2 = (d->tile & 255) != 1 && !stem && d->Wo >= 32 && d->Cin % tw == 0 &&
                    (long long)d->N * d->H * d->W * d->ldx * 4 < (1ll << 31) && (long long)a.M * d->ldy * 4 < (1ll << 31);
    if (v2) {
        // Every block asks for at least 64 KB of LDS, i.e. at most two weight-gradient blocks per CU: the 64 x 64 kernel
        // (32 KB) would otherwise fill a CU's 160 KB five deep and the input-gradient convs on the other stream (64 KB
        // tiles) would wait for a block to retire.  Training step 17.13 -> 17.01 ms; 96 KB (one block per CU): 17.67.
        static const size_t lds_min = getenv("PEMP_WGRAD_LDS") ? (size_t)atol(getenv("PEMP_WGRAD_LDS")) : 64 * 1024;
        size_t lds = big ? 2 * 2 * 32 * 32 * sizeof(v4f) : 2 * 2 * 32 * 16 * sizeof(v4f);
        if (lds < lds_min) lds = lds_min;
        if (lds > 64 * 1024) {
            if (big) (void)hipFuncSetAttribute((const void*)conv_wgrad2_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            else (void)hipFuncSetAttribute((const void*)conv_wgrad2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        static const bool xcd_off = getenv("PEMP_WGRAD_XCD") && getenv("PEMP_WGRAD_XCD")[0] == '0';
        a.xcd = xcd_off ? 0 : 1;
        a.nsplit_grid = grid.z;
        a.gx = grid.x;
        a.gy = grid.y;
        const dim3 grid1(grid.x * grid.y * grid.z);
        if (big) hipLaunchKernelGGL(conv_wgrad2_kernel<128>, grid1, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(conv_wgrad2_kernel<64>, grid1, dim3(256), lds, st, a);
    } else if (big) {
        hipLaunchKernelGGL(conv_wgrad128_kernel, grid, dim3(256), 2 * 2 * 32 * 32 * sizeof(v4f), st, a);
    } else {
        const size_t lds = 2 * 2 * 32 * 16 * sizeof(v4f);
        if (stem) hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(256), lds, st, a);
    }
    int e = launch_status("conv_wgrad");
    if (e || direct) return e;
    const long long n = (long long)d->Cout * d->Kpad;
    int rg = (int)std::min<long long>((n / 4 + 255) / 256, 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rg), dim3(256), 0, st, (const float*)ws, dw, n, a.nsplit, accumulate);
    return launch_status("conv_wgrad/reduce");
}
