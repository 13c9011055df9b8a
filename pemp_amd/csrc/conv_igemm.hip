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
#include "common.h"

namespace pemp {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ConvArgs {
    const float* x;
    const float* w;
    float* y;
    const float* scale;
    const float* shift;
    const float* res;
    int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, KH, KW, stride, pad, dil, ldr, Kpad;
    unsigned flags;
    int M, HoWo, cin_steps, nk, ntaps;
};

template <int BM, int BN, int WGM, bool STEM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AL = BM / 32, BL = BN / 32;  // float4 loads per thread per K step
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* As = smem;                    // [2][8][BM]
    float4* Bs = smem + 2 * 8 * BM;       // [2][8][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM;
    const int wn0 = (wave % WGN) * WN;
    const int lr = lane & 31, lh = lane >> 5;

    // n tiles vary fastest so that consecutive blocks reuse one A row-panel from L2
    const int ntn = a.Cout / BN;
    const int bm = blockIdx.x / ntn;
    const int bn = blockIdx.x % ntn;
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

    float4 ra[AL], rb[BL];

    auto gload = [&](int kt) {
        if constexpr (STEM) {
            int tap = kt * 8 + q;
            int kh = tap / a.KW, kw = tap - kh * a.KW;
            bool tok = tap < a.ntaps;
            int dh = kh * a.dil, dw = kw * a.dil;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
                bool ok = tok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                const float4* p = (const float4*)(a.x + (ptrdiff_t)(a_pix[i] + dh * a.W + dw) * 4);
                ra[i] = ok ? *p : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            int tap = kt / a.cin_steps;
            int cb = kt - tap * a.cin_steps;
            int kh = tap / a.KW, kw = tap - kh * a.KW;
            int dh = kh * a.dil, dw = kw * a.dil;
            int coff = cb * 32 + q * 4;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
                bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                const float4* p =
                    (const float4*)(a.x + (ptrdiff_t)(a_pix[i] + dh * a.W + dw) * a.ldx + coff);
                ra[i] = ok ? *p : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = *(const float4*)(wrow[i] + kt * 32);
    };

    auto lstore = [&](int buf) {
        float4* Ab = As + buf * 8 * BM + q * BM;
        float4* Bb = Bs + buf * 8 * BN + q * BN;
#pragma unroll
        for (int i = 0; i < AL; ++i) Ab[(r + 32 * i) ^ q] = ra[i];
#pragma unroll
        for (int i = 0; i < BL; ++i) Bb[(r + 32 * i) ^ q] = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < a.nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < a.nk;
        if (more) gload(kt + 1);

        const float4* Ab = As + buf * 8 * BM;
        const float4* Bb = Bs + buf * 8 * BN;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = 2 * j + lh;
            float4 af[TM], bf[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[mi] = Ab[qq * BM + ((wm0 + mi * 32 + lr) ^ qq)];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[ni] = Bb[qq * BN + ((wn0 + ni * 32 + lr) ^ qq)];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[i][j]: j = lane&31 (channel), i = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel) ----
    const bool relu = a.flags & PEMP_CONV_RELU;
    const bool per_img = a.flags & PEMP_CONV_SHIFT_PER_IMAGE;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + wn0 + ni * 32 + lr;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sh = (a.shift && !per_img) ? a.shift[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < a.M) {
                    float v = acc[mi][ni][e] * sc + sh;
                    if (per_img) v += a.shift[(size_t)(m / a.HoWo) * a.Cout + n];
                    if (a.res) v += a.res[(size_t)m * a.ldr + n];
                    if (relu) v = fmaxf(v, 0.f);
                    a.y[(size_t)m * a.ldy + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, bool STEM>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * (BM + BN) * sizeof(float4);
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
    PEMP_REQUIRE(d && x && w && y, "conv2d: null pointer");
    PEMP_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv2d: bad dims");
    PEMP_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "conv2d: bad kernel geometry");
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    PEMP_REQUIRE(ho == d->Ho && wo == d->Wo, "conv2d: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
    PEMP_REQUIRE(d->Cout % 64 == 0, "conv2d: Cout=%d must be a multiple of 64", d->Cout);
    PEMP_REQUIRE(d->ldy >= d->Cout, "conv2d: ldy < Cout");
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

    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.scale = scale; a.shift = shift; a.res = residual;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.ldx = d->ldx; a.Ho = d->Ho; a.Wo = d->Wo;
    a.Cout = d->Cout; a.ldy = d->ldy; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.dil = d->dil; a.ldr = d->ldr; a.Kpad = d->Kpad; a.flags = d->flags;
    a.HoWo = d->Ho * d->Wo;
    a.M = d->N * a.HoWo;
    a.ntaps = ntaps;
    a.cin_steps = stem ? 1 : d->Cin / 32;
    a.nk = d->Kpad / 32;
    if (residual) PEMP_REQUIRE(d->ldr >= d->Cout, "conv2d: ldr < Cout");

    int tile = d->tile;
    if (tile == 0) {
        // largest tile that still yields >= ~1.25 blocks per CU; else the small tile
        const long long b128 = (long long)cdiv(a.M, 128) * (a.Cout / 128 > 0 ? a.Cout / 128 : 0);
        const long long b12864 = (long long)cdiv(a.M, 128) * (a.Cout / 64);
        if (a.Cout % 128 == 0 && b128 >= 320) tile = 1;
        else if (b12864 >= 320) tile = 2;
        else tile = 3;
    }
    hipStream_t st = (hipStream_t)stream;
    if (tile == 1) {
        PEMP_REQUIRE(a.Cout % 128 == 0, "conv2d: tile 128x128 needs Cout %% 128 == 0");
        return stem ? launch_conv<128, 128, 2, true>(a, st) : launch_conv<128, 128, 2, false>(a, st);
    }
    if (tile == 2) return stem ? launch_conv<128, 64, 2, true>(a, st) : launch_conv<128, 64, 2, false>(a, st);
    if (tile == 3) return stem ? launch_conv<64, 64, 2, true>(a, st) : launch_conv<64, 64, 2, false>(a, st);
    set_error("conv2d: unknown tile id %d", tile);
    return -1;
}
