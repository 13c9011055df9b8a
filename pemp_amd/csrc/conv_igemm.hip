// Implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulate: bit-for-bit an fmaf chain, so parity with the fp32 reference is limited only by
// summation order).
//
//   GEMM view:  D[m][n] = sum_k A[m][k] * B[k][n]
//     m = (image, ho, wo) output pixel        M = N*Ho*Wo
//     n = output channel                      (Cout)
//     k = (kh, kw, ci), ci fastest            K = KH*KW*Cin      (NHWC input, KRSC weights)
//
//   Block = 256 threads = 4 waves, block tile BM x BN, K step 32 floats (= 8 "quads" of 4).
//   Each wave owns a (BM/WGM) x (BN/WGN) sub-tile made of 32x32 MFMA tiles.
//
//   LDS image (double buffered): A as float4 As[8 quads][BM rows], B as float4 Bs[8][BN];
//   element [q][r] holds k = 4q..4q+3 of row r and is stored at [q][r ^ q]: the 8 lanes that
//   stage one row's 128 contiguous bytes (q = 0..7) land on 8 different 16-B slots
//   (ds_write_b128 conflict-free), and a wave's ds_read_b128 of 32 consecutive rows of one quad
//   still covers 32 distinct slots.
//
//   MFMA operand order: lane l = (h = l>>5, r = l&31) reads quad 2j+h of row r and feeds its 4
//   floats to 4 consecutive MFMAs, i.e. MFMA e of pair j contracts k = {8j+e, 8j+4+e}.  A and B
//   use the same pairing, so the result is the plain dot product in a fixed, deterministic
//   order.
//
//   Staging: global -> registers (issued before the MFMAs of the current step) -> LDS buffer
//   b^1 after them; one __syncthreads per K step.
#include "conv_common.h"

namespace pemp {


template <int BM, int BN, int WGM, bool STEM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AL = BM / 32, BL = BN / 32;  // v4f loads per thread per K step
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) v4f smem[];
    v4f* As = smem;                    // [2][8][BM]
    v4f* Bs = smem + 2 * 8 * BM;       // [2][8][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM;
    const int wn0 = (wave % WGN) * WN;
    const int lr = lane & 31, lh = lane >> 5;

    // n tiles vary fastest so that consecutive blocks reuse one A row-panel from L2
    const int ntn = a.Cout / BN;
    const int tile_id = xcd_tile_order(blockIdx.x, gridDim.x);
    const int bm = tile_id / ntn;
    const int bn = tile_id % ntn;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- staging roles: thread (q, r) moves quad q of rows r + 32 i --------------------------
    const int q = tid & 7;
    const int r = tid >> 3;

    int a_pix[AL];   // element offset of (image, hi0, wi0) *without* ldx scaling: img*H*W + hi0*W + wi0
    int a_hi0[AL], a_wi0[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m = m0 + r + 32 * i;
        bool ok = m < a.M;
        int mm = ok ? m : 0;
        int img = mm / a.HoWo;
        int rem = mm - img * a.HoWo;
        int ho = rem / a.Wo;
        int wo = rem - ho * a.Wo;
        int hi0 = ho * a.stride - a.pad;
        int wi0 = wo * a.stride - a.pad;
        a_hi0[i] = ok ? hi0 : -(1 << 28);   // invalid rows fail every bounds test
        a_wi0[i] = wi0;
        a_pix[i] = img * a.H * a.W + hi0 * a.W + wi0;
    }
    const float* wrow[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) wrow[i] = a.w + (size_t)(n0 + r + 32 * i) * a.Kpad + q * 4;

    v4f ra[AL], rb[BL];
    unsigned okmask = 0;   // bit i: ra[i] holds real data (else it is zeroed before staging)

    // Source cursor of the generic path: K steps walk (tap outer, 32-channel chunk inner).  Inside
    // a tap every row pointer just advances by 32 floats; only a tap change re-derives pointers and
    // bounds (that keeps the per-step VALU/SALU work next to nothing).  Out-of-image taps point at
    // the tensor base (a harmless valid address) and are zeroed at staging time, so no load is
    // conditional per lane and the wave never waits on memory before its MFMAs.
    const float* pa[AL];
    const float* pb[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) pb[i] = wrow[i];
    int cur_tap = 0, cur_cb = 0;
#define PEMP_SET_TAP(tap_, cb_)                                                                      \
    do {                                                                                             \
        const int tap__ = (tap_);                                                                    \
        const int coff__ = (cb_) * 32 + q * 4;                                                       \
        const int kh = tap__ / a.KW, kw = tap__ - kh * a.KW;                                         \
        const int dh = kh * a.dil, dw = kw * a.dil;                                                  \
        const bool tok = tap__ < a.ntaps;                                                            \
        const bool pv__ = a.padv && tok;              /* out-of-image taps read the padding vector */ \
        const float* const oob__ = pv__ ? a.padv + coff__ : a.x;                                     \
        okmask = 0;                                                                                  \
        _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                             \
            const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;                                        \
            const bool ok = tok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;     \
            pa[i] = ok ? a.x + (ptrdiff_t)(a_pix[i] + dh * a.W + dw) * a.ldx + coff__ : oob__;       \
            okmask |= ((ok || pv__) ? 1u : 0u) << i;                                                 \
        }                                                                                            \
    } while (0)

    // loads of the CURRENT cursor position (generic) / of K step kt_ (stem)
#define PEMP_GLOAD(kt_)                                                                              \
    do {                                                                                             \
        if constexpr (STEM) {                                                                        \
            const int tap = (kt_) * 8 + q;                                                           \
            const int kh = tap / a.KW, kw = tap - kh * a.KW;                                         \
            const bool tok = tap < a.ntaps;                                                          \
            const int dh = kh * a.dil, dw = kw * a.dil;                                              \
            okmask = 0;                                                                              \
            _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                         \
                const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;                                    \
                const bool ok = tok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W; \
                const float* p = ok ? a.x + (ptrdiff_t)(a_pix[i] + dh * a.W + dw) * 4 : a.x;         \
                ra[i] = *(const v4f*)p;                                                              \
                okmask |= (ok ? 1u : 0u) << i;                                                       \
            }                                                                                        \
        } else {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < AL; ++i) ra[i] = *(const v4f*)pa[i];              \
        }                                                                                            \
        _Pragma("unroll") for (int i = 0; i < BL; ++i) rb[i] = *(const v4f*)pb[i];                   \
    } while (0)

    // move the cursor one K step forward
#define PEMP_ADVANCE()                                                                               \
    do {                                                                                             \
        if constexpr (STEM) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < BL; ++i) pb[i] += 32;                              \
        } else if (a.ntaps > 1) {                                                                    \
            /* multi-tap convs: channel chunk outer, tap inner (same order as conv_dma.hip: the taps \
               of one chunk re-read the same lines back to back -> L2 hits) */                       \
            if (++cur_tap == a.ntaps) {                                                              \
                cur_tap = 0;                                                                         \
                ++cur_cb;                                                                            \
            }                                                                                        \
            PEMP_SET_TAP(cur_tap, cur_cb);                                                           \
            const int koff = cur_tap * a.Cin + cur_cb * 32;                                          \
            _Pragma("unroll") for (int i = 0; i < BL; ++i) pb[i] = wrow[i] + koff;                   \
        } else {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < BL; ++i) pb[i] += 32;                              \
            _Pragma("unroll") for (int i = 0; i < AL; ++i) pa[i] += 32;                              \
        }                                                                                            \
    } while (0)

#define PEMP_LSTORE(buf_)                                                                            \
    do {                                                                                             \
        v4f* Ab_ = As + (buf_) * 8 * BM + q * BM;                                                 \
        v4f* Bb_ = Bs + (buf_) * 8 * BN + q * BN;                                                 \
        _Pragma("unroll") for (int i = 0; i < AL; ++i) {                                             \
            v4f v = ra[i];                                                                        \
            if (!((okmask >> i) & 1u)) v = v4f{0.f, 0.f, 0.f, 0.f};                          \
            Ab_[(r + 32 * i) ^ q] = v;                                                               \
        }                                                                                            \
        _Pragma("unroll") for (int i = 0; i < BL; ++i) Bb_[(r + 32 * i) ^ q] = rb[i];                \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    if constexpr (!STEM) PEMP_SET_TAP(0, 0);
    PEMP_GLOAD(0);
    PEMP_LSTORE(0);
    __syncthreads();

    const int arow = wm0 + lr, brow = wn0 + lr;
#ifndef PEMP_ABLATE
#define PEMP_ABLATE 0   // 1: no global loads in the loop, 2: + no LDS staging/barrier, 3: + no LDS reads (timing builds only)
#endif
    for (int kt = 0; kt < a.nk; ++kt) {
        const int buf = (PEMP_ABLATE >= 2) ? 0 : (kt & 1);
        if (PEMP_ABLATE < 1 && kt + 1 < a.nk) {   // wave-uniform branch; the last step re-stages stale registers (unused)
            PEMP_ADVANCE();
            PEMP_GLOAD(kt + 1);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMAs (hipcc sinks them to their use)

        const v4f* Ab = As + buf * 8 * BM;
        const v4f* Bb = Bs + buf * 8 * BN;
        v4f af[2][TM], bf[2][TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af[0][mi] = Ab[lh * BM + ((arow + mi * 32) ^ lh)];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) bf[0][ni] = Bb[lh * BN + ((brow + ni * 32) ^ lh)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) {   // prefetch the next quad pair while this one is in the matrix pipe
                const int qq = 2 * (j + 1) + lh;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[(j + 1) & 1][mi] = Ab[qq * BM + ((arow + mi * 32) ^ qq)];
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) bf[(j + 1) & 1][ni] = Bb[qq * BN + ((brow + ni * 32) ^ qq)];
            }
            __builtin_amdgcn_sched_barrier(0);   // reads of pair j+1 stay in front of the MFMAs of pair j
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
        if (PEMP_ABLATE < 2) {
            PEMP_LSTORE(buf ^ 1);
            __syncthreads();
        }
    }
#undef PEMP_GLOAD
#undef PEMP_ADVANCE
#undef PEMP_SET_TAP
#undef PEMP_LSTORE

    // ---- epilogue: transpose through LDS (staging buffers are free after the loop's last barrier) ----
    conv_epilogue_lds<TM, TN>(a, acc, (float*)smem + wave * 1024, m0 + wm0, n0 + wn0, lane);
}

template <int BM, int BN, int WGM, bool STEM>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(v4f);
    auto kern = conv_igemm_kernel<BM, BN, WGM, STEM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    const int grid = cdiv(a.M, BM) * (a.Cout / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return launch_status("conv_igemm");
}

}  // namespace pemp

using namespace pemp;

extern "C" int pemp_conv2d_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y,
                                    const float* scale, const float* shift, const float* residual,
                                    void* stream) {
    return pemp_conv2d_padv_nhwc_f32(d, x, w, y, scale, shift, residual, nullptr, stream);
}

static int conv2d_impl(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* scale, const float* shift,
                       const float* residual, const float* pad_value, void* ws, size_t ws_bytes, void* stream);

extern "C" int pemp_conv2d_padv_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y,
                                         const float* scale, const float* shift, const float* residual,
                                         const float* pad_value, void* stream) {
    PEMP_REQUIRE(!d || d->tile < 31 || d->tile > 37, "conv2d: the split-K tile ids 31..37 need pemp_conv2d_splitk_nhwc_f32 (workspace)");
    return conv2d_impl(d, x, w, y, scale, shift, residual, pad_value, nullptr, 0, stream);
}

extern "C" int pemp_conv2d_splitk_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y,
                                           const float* scale, const float* shift, const float* residual, void* ws,
                                           size_t ws_bytes, void* stream) {
    return conv2d_impl(d, x, w, y, scale, shift, residual, nullptr, ws, ws_bytes, stream);
}

extern "C" int pemp_conv2d_padv_splitk_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y,
                                                const float* scale, const float* shift, const float* residual,
                                                const float* pad_value, void* ws, size_t ws_bytes, void* stream) {
    PEMP_REQUIRE(pad_value, "conv2d_padv_splitk: pad_value is null");
    return conv2d_impl(d, x, w, y, scale, shift, residual, pad_value, ws, ws_bytes, stream);
}

// argument checks of one conv + its ConvArgs (``any_taps``: a padding value may accompany a 1x1 conv -- it is never read)
static int conv_fill(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* scale, const float* shift,
                     const float* residual, const float* pad_value, ConvArgs& a, bool any_taps = false) {
    PEMP_REQUIRE(d && x && w && y, "conv2d: null pointer");
    PEMP_REQUIRE(!pad_value || (!(d->flags & PEMP_CONV_STEM4) && (any_taps || d->KH * d->KW > 1) && ((uintptr_t)pad_value & 15) == 0),
                 "conv2d: pad_value needs a multi-tap non-stem conv and a 16-byte aligned [Cin] vector");
    PEMP_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv2d: bad dims");
    PEMP_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "conv2d: bad kernel geometry");
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    PEMP_REQUIRE(ho == d->Ho && wo == d->Wo, "conv2d: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
    PEMP_REQUIRE(d->Cout % 64 == 0, "conv2d: Cout=%d must be a multiple of 64", d->Cout);
    PEMP_REQUIRE(d->ldy >= d->Cout && d->ldy % 4 == 0 && ((uintptr_t)y & 15) == 0, "conv2d: ldy must be >= Cout and a multiple of 4, y 16-byte aligned");
    PEMP_REQUIRE(!scale || ((uintptr_t)scale & 15) == 0, "conv2d: scale must be 16-byte aligned");
    PEMP_REQUIRE(!shift || ((uintptr_t)shift & 15) == 0, "conv2d: shift must be 16-byte aligned");
    PEMP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv2d: x/w must be 16-byte aligned");
    const bool stem = d->flags & PEMP_CONV_STEM4;
    const int ntaps = d->KH * d->KW;
    if (stem) {
        PEMP_REQUIRE(d->Cin == 4 && d->ldx == 4, "conv2d: STEM4 needs NHWC4 input (Cin=ldx=4)");
        PEMP_REQUIRE(d->Kpad % 32 == 0 && d->Kpad >= ntaps * 4, "conv2d: STEM4 Kpad=%d too small / not x32", d->Kpad);
    } else {
        PEMP_REQUIRE(d->Cin % 32 == 0, "conv2d: Cin=%d must be a multiple of 32 (or use STEM4)", d->Cin);
        PEMP_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0, "conv2d: ldx=%d must be >= Cin and a multiple of 4", d->ldx);
        PEMP_REQUIRE(d->Kpad == ntaps * d->Cin, "conv2d: Kpad=%d must equal KH*KW*Cin=%d", d->Kpad, ntaps * d->Cin);
    }
    const long long in_elems = (long long)d->N * d->H * d->W * d->ldx;
    const long long out_elems = (long long)d->N * d->Ho * d->Wo * (long long)(d->ldy > d->ldr ? d->ldy : d->ldr);
    PEMP_REQUIRE(in_elems < (1ll << 31) && out_elems < (1ll << 31), "conv2d: tensor too large for 32-bit indexing");

    a.x = x; a.w = w; a.y = y; a.scale = scale; a.shift = shift; a.res = residual; a.padv = pad_value; a.stats = nullptr; a.bz = nullptr;
    a.bmask = nullptr; a.bmean = nullptr; a.binvstd = nullptr; a.ldbz = 0;
    a.rowmask = nullptr; a.rowcnt = nullptr;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Ho = d->Ho; a.Wo = d->Wo;
    a.Cout = d->Cout; a.ldy = d->ldy; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.dil = d->dil; a.ldr = d->ldr; a.Kpad = d->Kpad; a.flags = d->flags;
    a.HoWo = d->Ho * d->Wo;
    a.M = d->N * a.HoWo;
    a.ntaps = ntaps;
    a.cin_steps = stem ? 1 : d->Cin / 32;
    a.nk = d->Kpad / 32;
    if (residual) PEMP_REQUIRE(d->ldr >= d->Cout && d->ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0, "conv2d: ldr must be >= Cout and x4, residual 16-byte aligned");
    a.sk_ws = nullptr; a.sk_cnt = nullptr; a.sk_full = 0; a.sk_S = 1;
    a.bm_first = 0;
    return 0;
}

static int conv2d_impl(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* scale, const float* shift,
                       const float* residual, const float* pad_value, void* ws, size_t ws_bytes, void* stream) {
    ConvArgs a;
    const int rc = conv_fill(d, x, w, y, scale, shift, residual, pad_value, a);
    if (rc) return rc;
    const bool stem = d->flags & PEMP_CONV_STEM4;
    int tile = d->tile;
    if (tile == 0) {
        // Measured on MI355X (scratch/conv_tune.py): at these problem sizes (M <= ~80k rows) the 64x64
        // tile wins or ties everywhere -- waves per SIMD matter more than operand reuse for the
        // 64-cycle fp32 MFMA.  Callers that know better (engine autotune) pass an explicit tile.
        tile = 3;
    }
    hipStream_t st = (hipStream_t)stream;
    if (tile >= 31 && tile <= 37) {      // conv_dma2.hip with the last round of tiles split along K (pemp_hip.h)
        PEMP_REQUIRE(tile != 33, "conv2d: no split-K variant of the 64 x 64 tile");
        if (conv_dma2_supported(a)) {
            const int t = tile - 30;
            PEMP_REQUIRE((t != 1 && t != 4 && t != 6) || a.Cout % 128 == 0, "conv2d: tile N=128 needs Cout %% 128 == 0");
            PEMP_REQUIRE(t != 7 || a.Cout % 256 == 0, "conv2d: tile 256x256 needs Cout %% 256 == 0");
            return launch_conv_dma2_splitk(t, a, ws, ws_bytes, st);
        }
        tile -= 20;
    }
    if (tile == 29) {                    // hybrid (conv_dma2.hip): whole rounds of 32 x 32 wave tiles + the remaining rows on 16-row tiles, one grid
        if (conv_dma2_supported(a)) {
            const int rc2 = launch_conv_dma2_hybrid(a, st);
            if (rc2 != -2) return rc2;
        }
        tile = 23;                       // no hybrid split for this geometry: the 64 x 64 tile (same results)
    }
    if (tile >= 21 && tile <= 28) {      // conv_dma2.hip: buffer-addressed LDS-DMA + barrier inside the MFMA stream (same tile shapes as 11..17;
                                         // 28: the 16-row variant, 32 x 64 blocks on v_mfma_f32_16x16x4_f32)
        if (conv_dma2_supported(a)) {
            const int t = tile - 20;
            PEMP_REQUIRE((t != 1 && t != 4 && t != 6) || a.Cout % 128 == 0, "conv2d: tile N=128 needs Cout %% 128 == 0");
            PEMP_REQUIRE(t != 7 || a.Cout % 256 == 0, "conv2d: tile 256x256 needs Cout %% 256 == 0");
            return launch_conv_dma2(t, a, st);
        }
        tile = tile == 28 ? 13 : tile - 10;      // stem / padding value / > 32 taps / >= 2 GiB operands: the pointer-addressed variant
    }
    if (tile >= 11 && tile <= 17) {      // LDS-DMA staging variants (conv_dma.hip); 14..17: 8-wave blocks, 16: 256x128, 17: 256x256
        PEMP_REQUIRE((tile != 11 && tile != 14 && tile != 16) || a.Cout % 128 == 0, "conv2d: tile N=128 needs Cout %% 128 == 0");
        PEMP_REQUIRE(tile != 17 || a.Cout % 256 == 0, "conv2d: tile 256x256 needs Cout %% 256 == 0");
        return launch_conv_dma(tile - 10, a, st);
    }
    if (tile == 1) {
        PEMP_REQUIRE(a.Cout % 128 == 0, "conv2d: tile 128x128 needs Cout %% 128 == 0");
        return stem ? launch_conv<128, 128, 2, true>(a, st) : launch_conv<128, 128, 2, false>(a, st);
    }
    if (tile == 2) return stem ? launch_conv<128, 64, 2, true>(a, st) : launch_conv<128, 64, 2, false>(a, st);
    if (tile == 3) return stem ? launch_conv<64, 64, 2, true>(a, st) : launch_conv<64, 64, 2, false>(a, st);
    set_error("conv2d: unknown tile id %d", tile);
    return -1;
}


extern "C" int pemp_conv2d_group_nhwc_f32(int n, const pemp_conv_desc* d, const float* const* x, const float* const* w,
                                          float* const* y, const float* const* scale, const float* const* shift,
                                          const float* const* residual, const float* const* pad_value, void* stream) {
    PEMP_REQUIRE(n >= 1 && n <= CONV_GROUP_MAX && d && x && w && y, "conv2d_group: 1..%d member convs", CONV_GROUP_MAX);
    ConvGroupArgs g;
    g.n = n;
    const int tile = d[0].tile;
    PEMP_REQUIRE(tile >= 21 && tile <= 28, "conv2d_group: tile must be one of the buffer-addressed variants 21..28, got %d", tile);
    const int t = tile - 20;
    for (int i = 0; i < n; ++i) {
        PEMP_REQUIRE(d[i].tile == tile, "conv2d_group: every member must name the same tile variant");
        PEMP_REQUIRE(!pad_value || (pad_value[i] != nullptr) == (pad_value[0] != nullptr), "conv2d_group: pad_value for every member or for none");
        const int rc = conv_fill(&d[i], x[i], w[i], y[i], scale ? scale[i] : nullptr, shift ? shift[i] : nullptr,
                                 residual ? residual[i] : nullptr, pad_value ? pad_value[i] : nullptr, g.a[i], true);
        if (rc) return rc;
        PEMP_REQUIRE(conv_dma2_supported(g.a[i]), "conv2d_group: member %d lies outside the buffer-addressed kernels (stem / > 32 taps / 2 GiB operands / padding vector not behind the activations)", i);
        PEMP_REQUIRE((t != 1 && t != 4 && t != 6) || g.a[i].Cout % 128 == 0, "conv2d_group: tile N=128 needs Cout %% 128 == 0");
        PEMP_REQUIRE(t != 7 || g.a[i].Cout % 256 == 0, "conv2d_group: tile 256x256 needs Cout %% 256 == 0");
    }
    // members run beside each other in one grid: no member may write what another one writes or reads.  Two tensors that interleave
    // in one buffer (the ASPP branches write channel slices of the concat buffer: equal per-pixel stride, disjoint channel windows)
    // are fine; any other overlap of the byte ranges is refused.
    auto clash = [](const float* pa, long long rows_a, int ld_a, int c_a, const float* pb, long long rows_b, int ld_b, int c_b) {
        if (!pa || !pb) return false;
        const char *a0 = (const char*)pa, *a1 = a0 + ((rows_a - 1) * ld_a + c_a) * 4;
        const char *b0 = (const char*)pb, *b1 = b0 + ((rows_b - 1) * ld_b + c_b) * 4;
        if (a1 <= b0 || b1 <= a0) return false;                       // disjoint ranges
        if (ld_a != ld_b) return true;
        const long long delta = (b0 - a0) / 4, r = ((delta % ld_a) + ld_a) % ld_a;      // b's window inside a's pixel
        return !(r >= c_a && r + c_b <= ld_a);
    };
    for (int i = 0; i < n; ++i) {
        const long long Mi = (long long)g.a[i].N * g.a[i].Ho * g.a[i].Wo;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            const long long Mj = (long long)g.a[j].N * g.a[j].Ho * g.a[j].Wo, Pj = (long long)g.a[j].N * g.a[j].H * g.a[j].W;
            PEMP_REQUIRE(j > i || !clash(g.a[i].y, Mi, g.a[i].ldy, g.a[i].Cout, g.a[j].y, Mj, g.a[j].ldy, g.a[j].Cout),
                         "conv2d_group: members %d and %d write the same output", i, j);
            PEMP_REQUIRE(!clash(g.a[i].y, Mi, g.a[i].ldy, g.a[i].Cout, g.a[j].x, Pj, g.a[j].ldx, g.a[j].Cin),
                         "conv2d_group: member %d writes what member %d reads as its input", i, j);
            PEMP_REQUIRE(!clash(g.a[i].y, Mi, g.a[i].ldy, g.a[i].Cout, g.a[j].res, Mj, g.a[j].ldr, g.a[j].Cout),
                         "conv2d_group: member %d writes what member %d reads as its residual", i, j);
        }
    }
    for (int i = n; i < CONV_GROUP_MAX; ++i) g.a[i] = g.a[0];
    return launch_conv_dma2_group(t, g, (hipStream_t)stream);
}


// conv (+ affine, residual, ReLU) followed by DropBlock2D's scaling of the output rows, in one launch: the buffer-addressed
// kernels only (tiles 21..27, or 31..37 with a split-K workspace)
extern "C" int pemp_conv2d_dropblock_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* scale,
                                              const float* shift, const float* residual, const float* rowmask, const int* kept_count,
                                              void* ws, size_t ws_bytes, void* stream) {
    PEMP_REQUIRE(rowmask && kept_count, "conv2d_dropblock: null mask / count");
    ConvArgs a;
    const int rc = conv_fill(d, x, w, y, scale, shift, residual, nullptr, a);
    if (rc) return rc;
    PEMP_REQUIRE(d->tile >= 21 && d->tile <= 37 && d->tile != 33 && !(d->tile >= 28 && d->tile <= 30),
                 "conv2d_dropblock: tile must be 21..27 or 31..37 (no 33), got %d", d->tile);
    PEMP_REQUIRE(conv_dma2_supported(a), "conv2d_dropblock: geometry / operand size outside the buffer-addressed kernels");
    a.rowmask = rowmask;
    a.rowcnt = kept_count;
    const int t = d->tile > 30 ? d->tile - 30 : d->tile - 20;
    PEMP_REQUIRE((t != 1 && t != 4 && t != 6) || a.Cout % 128 == 0, "conv2d_dropblock: tile N=128 needs Cout %% 128 == 0");
    PEMP_REQUIRE(t != 7 || a.Cout % 256 == 0, "conv2d_dropblock: tile 256x256 needs Cout %% 256 == 0");
    PEMP_REQUIRE(d->tile != 33, "conv2d_dropblock: no split-K variant of the 64 x 64 tile");
    return launch_conv_dma2_db(t, a, ws, ws_bytes, d->tile > 30, (hipStream_t)stream);
}


// The side-figure variant: bf16 operands, fp32 accumulation (conv_dma2.hip, BF16).  The descriptor is in ELEMENTS like every
// other; the kernel is handed Cin / ldx / Kpad in dwords (two bf16 each), which is all it needs to address bf16 rows.
extern "C" int pemp_conv2d_bf16_nhwc(const pemp_conv_desc* d, const void* x, const void* w, void* y, const float* scale,
                                     const float* shift, const void* residual, const void* pad_value, int out_f32, void* stream) {
    PEMP_REQUIRE(d && x && w && y, "conv2d_bf16: null pointer");
    PEMP_REQUIRE(!(d->flags & (PEMP_CONV_STEM4 | PEMP_CONV_BF16_IO)), "conv2d_bf16: no stem variant; unknown flags");
    PEMP_REQUIRE(d->Cin % 64 == 0 && d->ldx % 8 == 0 && d->ldx >= d->Cin && d->Kpad == d->KH * d->KW * d->Cin,
                 "conv2d_bf16: Cin must be a multiple of 64, ldx of 8 (bf16 elements), Kpad = KH*KW*Cin");
    PEMP_REQUIRE(!residual || !out_f32, "conv2d_bf16: a residual comes with a bf16 output");
    PEMP_REQUIRE(d->tile == 0 || (d->tile >= 21 && d->tile <= 27), "conv2d_bf16: tile must be 0 or 21..27");
    pemp_conv_desc h = *d;
    h.Cin = d->Cin / 2;
    h.ldx = d->ldx / 2;
    h.Kpad = d->Kpad / 2;
    ConvArgs a;
    // (the fp32 checks apply to the halved descriptor: Cin % 32, ldx % 4, 16-byte aligned operands; y / residual strides % 4)
    const int rc = conv_fill(&h, (const float*)x, (const float*)w, (float*)y, scale, shift, (const float*)residual,
                             (const float*)pad_value, a);
    if (rc) return rc;
    if (!out_f32) a.flags |= PEMP_CONV_BF16_IO;
    PEMP_REQUIRE(conv_dma2_supported(a), "conv2d_bf16: geometry / operand size outside the buffer-addressed kernels");
    const int t = d->tile == 0 ? 4 : d->tile - 20;
    PEMP_REQUIRE((t != 1 && t != 4 && t != 6) || a.Cout % 128 == 0, "conv2d_bf16: tile N=128 needs Cout %% 128 == 0");
    PEMP_REQUIRE(t != 7 || a.Cout % 256 == 0, "conv2d_bf16: tile 256x256 needs Cout %% 256 == 0");
    return launch_conv_dma2_bf16(t, a, (hipStream_t)stream);
}


static int conv_stats_fill(const char* what, const pemp_conv_desc* d, ConvArgs& a) {
    PEMP_REQUIRE(!(d->flags & (PEMP_CONV_STEM4 | PEMP_CONV_RELU | PEMP_CONV_SHIFT_PER_IMAGE)), "%s: plain conv only (no stem / ReLU / per-image shift)", what);
    PEMP_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0,
                 "%s: bad geometry", what);
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    PEMP_REQUIRE(ho == d->Ho && wo == d->Wo, "%s: Ho/Wo do not match geometry", what);
    PEMP_REQUIRE(d->Cout % 64 == 0 && d->Cin % 32 == 0 && d->ldx >= d->Cin && d->ldx % 4 == 0 && d->ldy >= d->Cout && d->ldy % 4 == 0,
                 "%s: Cout %% 64, Cin %% 32, strides %% 4", what);
    PEMP_REQUIRE(d->Kpad == d->KH * d->KW * d->Cin, "%s: Kpad must equal KH*KW*Cin", what);
    a.scale = nullptr; a.shift = nullptr; a.padv = nullptr; a.rowmask = nullptr; a.rowcnt = nullptr;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Ho = d->Ho; a.Wo = d->Wo;
    a.Cout = d->Cout; a.ldy = d->ldy; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.dil = d->dil; a.ldr = d->ldr; a.Kpad = d->Kpad; a.flags = d->flags;
    a.HoWo = d->Ho * d->Wo;
    a.M = d->N * a.HoWo;
    a.ntaps = d->KH * d->KW;
    a.cin_steps = d->Cin / 32;
    a.nk = d->Kpad / 32;
    a.sk_ws = nullptr; a.sk_cnt = nullptr; a.sk_full = 0; a.sk_S = 1;
    a.bm_first = 0;
    if (!conv_dma2_supported(a)) {
        set_error("%s: geometry / operand size outside the buffer-addressed kernels", what);
        return -2;
    }
    const int t = d->tile == 0 ? 3 : (d->tile > 30 ? d->tile - 30 : d->tile - 20);
    PEMP_REQUIRE(t >= 1 && t <= 7 && d->tile != 33, "%s: tile must be 0, 21..27 or 31..37 (no 33)", what);
    PEMP_REQUIRE((t != 1 && t != 4 && t != 6) || a.Cout % 128 == 0, "%s: tile N=128 needs Cout %% 128 == 0", what);
    PEMP_REQUIRE(t != 7 || a.Cout % 256 == 0, "%s: tile 256x256 needs Cout %% 256 == 0", what);
    return 0;
}

static int conv_stats_common(const char* what, const pemp_conv_desc* d, ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    const int rc = conv_stats_fill(what, d, a);
    if (rc) return rc;
    if (d->tile > 30) return launch_conv_dma2_splitk(d->tile - 30, a, ws, ws_bytes, st);
    return launch_conv_dma2(d->tile == 0 ? 3 : d->tile - 20, a, st);
}

extern "C" int pemp_conv2d_hybrid_rows(const pemp_conv_desc* d) {
    if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return 0;
    return conv_dma2_hybrid_rows(d->N * d->Ho * d->Wo, d->Cout);
}

extern "C" int pemp_conv2d_stats_rows(const pemp_conv_desc* d) {
    if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
    const int t = d->tile == 0 ? 3 : (d->tile > 30 ? d->tile - 30 : d->tile - 20);
    if (t < 1 || t > 7) return 0;            // 28 / 29 (16-row and hybrid launches) have no statistics epilogue
    const int bm = conv_dma2_tile_rows(t);
    return bm ? cdiv(d->N * d->Ho * d->Wo, bm) : 0;
}

extern "C" size_t pemp_conv2d_splitk_workspace_bytes(const pemp_conv_desc* d) {
    if (!d || d->tile < 31 || d->tile > 37 || d->tile == 33 || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 || d->Kpad < 32) return 0;
    ConvArgs a;
    a.M = d->N * d->Ho * d->Wo;
    a.Cout = d->Cout;
    a.nk = d->Kpad / 32;
    return conv_dma2_splitk_plan(d->tile - 30, a).ws_bytes;
}

extern "C" int pemp_conv2d_stats_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, float* stats,
                                          void* ws, size_t ws_bytes, void* stream) {
    PEMP_REQUIRE(d && x && w && y && stats, "conv2d_stats: null pointer");
    PEMP_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)stats) & 15) == 0, "conv2d_stats: pointers must be 16-byte aligned");
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.res = nullptr; a.stats = stats;
    a.bmask = nullptr; a.bz = nullptr; a.bmean = nullptr; a.binvstd = nullptr; a.ldbz = 0;
    return conv_stats_common("conv2d_stats", d, a, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int pemp_conv2d_bnbwd_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* residual,
                                          const uint32_t* mask, const float* z, int ldz, const float* mean, const float* invstd,
                                          float* stats, void* ws, size_t ws_bytes, void* stream) {
    PEMP_REQUIRE(d && x && w && y && z && mean && invstd && stats, "conv2d_bnbwd: null pointer");
    PEMP_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)stats | (uintptr_t)z | (uintptr_t)residual | (uintptr_t)mean |
                   (uintptr_t)invstd) & 15) == 0 && ((uintptr_t)mask & 3) == 0, "conv2d_bnbwd: pointers must be 16-byte aligned");
    PEMP_REQUIRE(ldz >= d->Cout && ldz % 4 == 0 && (!residual || (d->ldr >= d->Cout && d->ldr % 4 == 0)), "conv2d_bnbwd: ldz / ldr");
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.res = residual; a.stats = stats;
    a.bmask = mask; a.bz = z; a.bmean = mean; a.binvstd = invstd; a.ldbz = ldz;
    return conv_stats_common("conv2d_bnbwd", d, a, ws, ws_bytes, (hipStream_t)stream);
}
