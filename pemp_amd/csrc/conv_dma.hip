// Implicit-GEMM convolution, LDS-DMA staging variant (global_load_lds_dwordx4): operand tiles go
// HBM/L2 -> LDS without passing through VGPRs, so the K loop has no staging registers and no
// ds_write at all.  Same GEMM view, same MFMA order and therefore bit-identical results to
// conv_igemm.hip; only the LDS image differs:
//
//   A tile  As[buf][row][8 quads] (row-major, 128 B per row; B tile alike).  One DMA
//   wave-instruction moves 8 rows x 128 B: lane (rl = lane>>3, p = lane&7) lands at
//   row*128 + p*16, i.e. the image is lane-linear as the hardware requires (dst = M0 + lane*16).
//   To keep the MFMA fragment reads (32 rows x one quad per ds_read_b128) conflict-free the quad
//   order inside a row is XOR-swizzled ON THE SOURCE SIDE: position p of row r holds quad
//   p ^ ((r>>1)&7); the lane simply fetches that quad of its row (same 128-B line, so coalescing
//   is untouched) and the reader looks quad Q up at position Q ^ ((r>>1)&7).
//
//   Out-of-image taps (zero padding) are fetched from a 16-byte zero block.
#include "conv_common.h"

namespace pemp {

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int WGM, bool STEM, int NW>
__global__ __launch_bounds__(NW * 64) void conv_dma_kernel(ConvArgs a) {
    constexpr int WGN = NW / WGM;
    constexpr int RPI = NW * 8;                 // rows covered by one DMA instruction round of the block
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AL = BM / RPI, BL = BN / RPI; // DMA wave-instructions per thread per K step
    constexpr int NBUF = 2;

    extern __shared__ __attribute__((aligned(16))) v4f smem[];
    v4f* As = smem;                      // [NBUF][BM][8]
    v4f* Bs = smem + NBUF * BM * 8;      // [NBUF][BN][8]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WGN) * WM;
    const int wn0 = (wave % WGN) * WN;
    const int lr = lane & 31, lh = lane >> 5;

    const int ntn = a.Cout / BN;
    const int tile_id = xcd_tile_order(blockIdx.x, gridDim.x);
    const int bm = tile_id / ntn;
    const int bn = tile_id % ntn;
    const int m0 = bm * BM, n0 = bn * BN;

    // loader role: thread (r, p) fetches, for rows r + 32 i, the quad that belongs at position p
    const int p = tid & 7;
    const int r = tid >> 3;
    const int sq = p ^ ((r >> 1) & 7);          // source quad (same for every i: RPI i >> 1 = 0 mod 8)
    const float* zero = g_zero16;

    int a_pix[AL], a_hi0[AL], a_wi0[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m = m0 + r + RPI * i;
        bool ok = m < a.M;
        int mm = ok ? m : 0;
        int img = mm / a.HoWo;
        int rem = mm - img * a.HoWo;
        int ho = rem / a.Wo;
        int wo = rem - ho * a.Wo;
        int hi0 = ho * a.stride - a.pad;
        int wi0 = wo * a.stride - a.pad;
        a_hi0[i] = ok ? hi0 : -(1 << 28);
        a_wi0[i] = wi0;
        a_pix[i] = img * a.H * a.W + hi0 * a.W + wi0;
    }
    const float* pa[AL];
    const float* pb[BL];
    const float* pb0[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) pb0[i] = pb[i] = a.w + (size_t)(n0 + r + RPI * i) * a.Kpad + sq * 4;
    int cur_tap = 0, cur_cb = 0;

    // K-step order of multi-tap convs: channel chunk OUTER, tap INNER -- the 9 taps of one 32-channel chunk re-read
    // almost the same cache lines back to back (L2 hits) instead of streaming the whole input once per tap.
    // Measured on the 3x3 256->256 layer (50 x 51 x 51, 256x256 tile): L2 fill traffic 1120 MB -> 166 MB per
    // launch at unchanged speed.  Every conv kernel uses this order, so variants stay bit-identical.
#define PEMP_SET_TAP(tap_, cb_)                                                                      \
    do {                                                                                             \
        const int tap__ = (tap_);                                                                    \
        const int coff__ = (cb_) * 32 + sq * 4;                                                      \
        const int kh = tap__ / a.KW, kw = tap__ - kh * a.KW;                                         \
        const int dh = kh * a.dil, dw = kw * a.dil;                                                  \
        const bool tok = tap__ < a.ntaps;                                                            \
        const float* const oob__ = (a.padv && tok) ? a.padv + coff__ : zero;   /* padding value of this channel quad */ \
        _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                             \
            const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;                                        \
            const bool ok = tok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;     \
            pa[i] = ok ? a.x + (ptrdiff_t)(a_pix[i] + dh * a.W + dw) * a.ldx + coff__ : oob__;       \
        }                                                                                            \
    } while (0)

    // issue the DMA of the cursor's K step (generic) / of K step kt_ (stem) into buffer buf_
#define PEMP_DMA(kt_, buf_)                                                                          \
    do {                                                                                             \
        v4f* Ad_ = As + (buf_) * BM * 8 + wave * 64;                                                 \
        v4f* Bd_ = Bs + (buf_) * BN * 8 + wave * 64;                                                 \
        if constexpr (STEM) {                                                                        \
            const int tap = (kt_) * 8 + sq;                                                          \
            const int kh = tap / a.KW, kw = tap - kh * a.KW;                                         \
            const bool tok = tap < a.ntaps;                                                          \
            const int dh = kh * a.dil, dw = kw * a.dil;                                              \
            _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                         \
                const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;                                    \
                const bool ok = tok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W; \
                const float* s_ = ok ? a.x + (ptrdiff_t)(a_pix[i] + dh * a.W + dw) * 4 : zero;       \
                __builtin_amdgcn_global_load_lds((gptr_t)s_, (lptr_t)(Ad_ + i * RPI * 8), 16, 0, 0); \
            }                                                                                        \
        } else {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < AL; ++i)                                           \
                __builtin_amdgcn_global_load_lds((gptr_t)pa[i], (lptr_t)(Ad_ + i * RPI * 8), 16, 0, 0); \
        }                                                                                            \
        _Pragma("unroll") for (int i = 0; i < BL; ++i)                                               \
            __builtin_amdgcn_global_load_lds((gptr_t)pb[i], (lptr_t)(Bd_ + i * RPI * 8), 16, 0, 0); \
    } while (0)

#define PEMP_ADVANCE()                                                                               \
    do {                                                                                             \
        if constexpr (STEM) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < BL; ++i) pb[i] += 32;                              \
        } else if (a.ntaps > 1) {                                                \
            if (++cur_tap == a.ntaps) {                                                              \
                cur_tap = 0;                                                                         \
                ++cur_cb;                                                                            \
            }                                                                                        \
            PEMP_SET_TAP(cur_tap, cur_cb);                                                           \
            const int koff = cur_tap * a.Cin + cur_cb * 32;                                          \
            _Pragma("unroll") for (int i = 0; i < BL; ++i) pb[i] = pb0[i] + koff;                    \
        } else {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < BL; ++i) pb[i] += 32;                              \
            if (++cur_cb == a.cin_steps) {                                                           \
                cur_cb = 0;                                                                          \
                ++cur_tap;                                                                           \
                PEMP_SET_TAP(cur_tap, 0);                                                            \
            } else {                                                                                 \
                _Pragma("unroll") for (int i = 0; i < AL; ++i) pa[i] = pa[i] == zero ? zero : pa[i] + 32; \
            }                                                                                        \
        }                                                                                            \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    if constexpr (!STEM) PEMP_SET_TAP(0, 0);
    PEMP_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment read positions: quad Q of row (.. + lr) sits at position Q ^ ((lr>>1)&7)
    const int rsw = (lr >> 1) & 7;
    const int arow = (wm0 + lr) * 8, brow = (wn0 + lr) * 8;
    for (int kt = 0; kt < a.nk; ++kt) {
        const int buf = kt & 1;
#ifndef PEMP_ABLATE
#define PEMP_ABLATE 0   // timing builds only: 1 = no DMA in the loop, 2 = DMA but no wait/barrier, 3 = neither
#endif
        if (kt + 1 < a.nk && !(PEMP_ABLATE & 1)) {      // wave-uniform
            PEMP_ADVANCE();
            PEMP_DMA(kt + 1, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);

        const v4f* Ab = As + buf * BM * 8;
        const v4f* Bb = Bs + buf * BN * 8;
        v4f af[2][TM], bf[2][TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af[0][mi] = Ab[arow + mi * 256 + (lh ^ rsw)];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) bf[0][ni] = Bb[brow + ni * 256 + (lh ^ rsw)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) {
                const int pos = (2 * (j + 1) + lh) ^ rsw;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[(j + 1) & 1][mi] = Ab[arow + mi * 256 + pos];
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) bf[(j + 1) & 1][ni] = Bb[brow + ni * 256 + pos];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const v4f av = af[j & 1][mi], bv = bf[j & 1][ni];
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[mi][ni], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(PEMP_ABLATE & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of step kt+1 have landed
            __syncthreads();                                    // ... and everybody's; all reads of `buf` are done
        }
    }
#undef PEMP_DMA
#undef PEMP_ADVANCE
#undef PEMP_SET_TAP

    // ---- epilogue: transpose through LDS (staging buffers are free after the loop's last barrier) ----
    conv_epilogue_lds<TM, TN>(a, acc, (float*)smem + wave * 1024, m0 + wm0, n0 + wn0, lane);
}

template <int BM, int BN, int WGM, bool STEM, int NW = 4>
static int launch_dma(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(v4f);
    auto kern = conv_dma_kernel<BM, BN, WGM, STEM, NW>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    const int grid = cdiv(a.M, BM) * (a.Cout / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    return launch_status("conv_dma");
}

int launch_conv_dma(int tile, const ConvArgs& a, hipStream_t st) {
    const bool stem = a.flags & PEMP_CONV_STEM4;
    if (tile == 7) return stem ? launch_dma<256, 256, 4, true, 8>(a, st) : launch_dma<256, 256, 4, false, 8>(a, st);
    if (tile == 6) return stem ? launch_dma<256, 128, 4, true, 8>(a, st) : launch_dma<256, 128, 4, false, 8>(a, st);
    if (tile == 4) return stem ? launch_dma<128, 128, 4, true, 8>(a, st) : launch_dma<128, 128, 4, false, 8>(a, st);
    if (tile == 5) return stem ? launch_dma<128, 64, 4, true, 8>(a, st) : launch_dma<128, 64, 4, false, 8>(a, st);
    if (tile == 1) return stem ? launch_dma<128, 128, 2, true>(a, st) : launch_dma<128, 128, 2, false>(a, st);
    if (tile == 2) return stem ? launch_dma<128, 64, 2, true>(a, st) : launch_dma<128, 64, 2, false>(a, st);
    return stem ? launch_dma<64, 64, 2, true>(a, st) : launch_dma<64, 64, 2, false>(a, st);
}

}  // namespace pemp
