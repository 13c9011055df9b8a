// Prototype head of PEMP on gfx950: meta-prototype module (soft assignment + masked pooling),
// plain / full-resolution masked average pooling, pixel x prototype cosine map with group max,
// and the upsample / argmax / cross-entropy / IoU-count tail of the evaluator.
//
// All of these are streaming kernels over [pixels][c] feature maps (c contiguous): their bound is HBM
// bandwidth (AI ~ 3 flop/B at 2p = 6 prototypes).  The three passes over the features run on the matrix
// cores -- per-pixel products with a c x 2p table as 16x16x4 fp32 MFMA row streams (cosine map, MPM
// assignment), masked pooling as an MFMA over pixels -- with lane-contiguous loads, 8-16 KB per wave in
// flight and at most 96 VGPRs / 32 KB of LDS per block so that five blocks share a CU (measured 5.4-5.7
// TB/s).  Wave-per-pixel VALU variants remain for channel counts the MFMA kernels are not instantiated
// for.  Partial sums are combined in a fixed order everywhere (deterministic).
#include <stdlib.h>
#include "head_common.h"

namespace pemp {

// -----------------------------------------------------------------------------------------------
// assign weights  A[bs][j][i]
//   MODE 0: meta-prototype soft assignment, networks/pemp_stage1.py:205-207
//   MODE 1: plain masks (fg, bg) at feature resolution, pemp_stage1.py:224-225
// one wave per pixel.
template <int MODE>
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ feat, int ldf,
                                                     const float* __restrict__ mask,
                                                     const float* __restrict__ ctr, float* __restrict__ A,
                                                     int n, int h, int w, int H, int W, int c, int p) {
    const int bs = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int J = MODE == 0 ? 2 * p : 2;
    const int ncl = (c + 255) / 256;  // float4 chunks per lane

    float cw[MAXJ][MAXCL];            // this lane's slice of ctr (MODE 0)
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int ch = t * 256 + lane * 4 + e;
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) cw[j][t * 4 + e] = (t < ncl && ch < c && j < J) ? ctr[ch * J + j] : 0.f;
            }
    }
    const float* mk = mask + (size_t)bs * 2 * H * W;
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        const int y = i / w, x = i - y * w;
        const int sy = nearest_src(y, H, h), sx = nearest_src(x, W, w);
        const float mfg = mk[(size_t)sy * W + sx];
        const float mbg = mk[(size_t)H * W + (size_t)sy * W + sx];
        float* out = A + ((size_t)bs * J) * n + i;
        if (MODE == 1) {
            if (lane == 0) {
                out[0] = mfg;
                out[n] = mbg;
            }
            continue;
        }
        const float* xp = feat + ((size_t)bs * n + i) * ldf;
        float d[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) d[j] = 0.f;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            if (t < ncl && ch < c) {
                float4 v = *(const float4*)(xp + ch);
                float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j) {
                        float df = xv[e] - cw[j][t * 4 + e];
                        d[j] += df * df;
                    }
            }
        }
        {
            const float tot = wave_sum8(d);          // lane l holds the total of d[l & 7]
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) d[j] = -__shfl(tot, j, 64);
        }
        if (lane == 0) {
            for (int g = 0; g < 2; ++g) {
                float mx = -INFINITY;
                for (int j = 0; j < p; ++j) mx = fmaxf(mx, d[g * p + j]);
                float e[MAXJ / 2], s = 0.f;
                for (int j = 0; j < p; ++j) {
                    e[j] = expf(d[g * p + j] - mx);
                    s += e[j];
                }
                const float m = g == 0 ? mfg : mbg;
                for (int j = 0; j < p; ++j) out[(size_t)(g * p + j) * n] = (e[j] / s) * m;
            }
        }
    }
}

// MODE 2 (Baseline): A_g[p] = sum_P m_g[P] * W[P][p], W = bilinear align_corners weights of the
// h x w -> H x W upsampling (adjoint of networks/baseline.py:100), plus exact mask sums.
// one thread per low-res pixel and group.
__global__ void adjoint_mask_kernel(const float* __restrict__ mask, float* __restrict__ A, int n, int h, int w,
                                    int H, int W) {
    const int bs = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * n) return;
    const int g = idx / n, i = idx - g * n;
    const int y = i / w, x = i - y * w;
    const float* m = mask + ((size_t)bs * 2 + g) * H * W;
    const float sh = h > 1 && H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = w > 1 && W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    // rows Y whose source coordinate sh*Y lies in (y-1, y+1)
    int Y0 = sh > 0.f ? max(0, (int)floorf((float)(y - 1) / sh) - 1) : 0;
    int Y1 = sh > 0.f ? min(H - 1, (int)ceilf((float)(y + 1) / sh) + 1) : H - 1;
    int X0 = sw > 0.f ? max(0, (int)floorf((float)(x - 1) / sw) - 1) : 0;
    int X1 = sw > 0.f ? min(W - 1, (int)ceilf((float)(x + 1) / sw) + 1) : W - 1;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        float fy = sh * (float)Y;
        int y0 = (int)fy;
        int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        float ly = fy - (float)y0;
        float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y && y1 != y0 ? ly : 0.f);
        if (y1 == y0 && y0 == y) wy = 1.f;
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = X0; X <= X1; ++X) {
            float fx = sw * (float)X;
            int x0 = (int)fx;
            int x1 = x0 + (x0 < w - 1 ? 1 : 0);
            float lx = fx - (float)x0;
            float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x && x1 != x0 ? lx : 0.f);
            if (x1 == x0 && x0 == x) wx = 1.f;
            if (wx != 0.f) row += wx * m[(size_t)Y * W + X];
        }
        acc += wy * row;
    }
    A[((size_t)bs * 2 + g) * n + i] = acc;
}

// exact full-resolution mask sums (integers in fp32, so the order is immaterial), one 1024-thread block per (bs, g)
__global__ __launch_bounds__(1024) void mask_sum_kernel(const float* __restrict__ mask, float* __restrict__ out,
                                                        int HW) {
    __shared__ float red[16];
    const float* m = mask + (size_t)blockIdx.x * HW;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = threadIdx.x;
    for (; i + 3072 < HW; i += 4096) {          // four independent loads in flight per thread
        s0 += m[i];
        s1 += m[i + 1024];
        s2 += m[i + 2048];
        s3 += m[i + 3072];
    }
    for (; i < HW; i += 1024) s0 += m[i];
    float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[k];
        out[blockIdx.x] = t;
    }
}

// -----------------------------------------------------------------------------------------------
// pooling partial sums: part[bs][chunk][j][c] = sum_{i in chunk} x[i][c] * A[j][i]
//                       asum[bs][chunk][j]    = sum_{i in chunk} A[j][i]
// thread t owns channels t and t+256 (coalesced across the block), pixels in order.
__global__ __launch_bounds__(256) void pool_partial_kernel(const float* __restrict__ feat, int ldf,
                                                           const float* __restrict__ A, float* __restrict__ part,
                                                           float* __restrict__ asum, int n, int c, int J,
                                                           int nchunks) {
    __shared__ float As[MAXJ][PCHUNK];
    const int bs = blockIdx.y, ck = blockIdx.x;
    const int i0 = ck * PCHUNK;
    const int np = min(PCHUNK, n - i0);
    for (int t = threadIdx.x; t < J * PCHUNK; t += 256) {
        int j = t / PCHUNK, i = t - j * PCHUNK;
        As[j][i] = i < np ? A[((size_t)bs * J + j) * n + i0 + i] : 0.f;
    }
    __syncthreads();
    float acc[2][MAXJ];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) acc[k][j] = 0.f;
    const float* xb = feat + ((size_t)bs * n + i0) * ldf;
    for (int i = 0; i < np; ++i) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int ch = threadIdx.x + 256 * k;
            if (ch < c) {
                float v = xb[(size_t)i * ldf + ch];
#pragma unroll
                for (int j = 0; j < MAXJ; ++j)
                    if (j < J) acc[k][j] += v * As[j][i];
            }
        }
    }
    float* pb = part + ((size_t)bs * nchunks + ck) * J * c;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int ch = threadIdx.x + 256 * k;
        if (ch < c)
#pragma unroll
            for (int j = 0; j < MAXJ; ++j)
                if (j < J) pb[(size_t)j * c + ch] = acc[k][j];
    }
    if (threadIdx.x < J) {
        float s = 0.f;
        for (int i = 0; i < np; ++i) s += As[threadIdx.x][i];
        asum[((size_t)bs * nchunks + ck) * J + threadIdx.x] = s;
    }
}

// protos[b][j][c] = mean_s ( sum_chunks part / (denominator + eps) ); block = 64 channels x 4 chunk lanes
__global__ __launch_bounds__(256) void pool_final_kernel(const float* __restrict__ part,
                                                         const float* __restrict__ asum,
                                                         const float* __restrict__ den_override,
                                                         float* __restrict__ protos, int S, int c, int J, int nchunks,
                                                         float eps) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, j = blockIdx.x;
    const int ch = blockIdx.z * 64 + (threadIdx.x & 63), chl = min(ch, c - 1);
    float tot = 0.f;
    for (int s = 0; s < S; ++s) {
        const int bs = b * S + s;
        const float num = chunk_sum(part + ((size_t)bs * nchunks * J + j) * c + chl, (size_t)J * c, nchunks, red);
        float den = chunk_sum(asum + (size_t)bs * nchunks * J + j, (size_t)J, nchunks, red);
        if (den_override) den = den_override[bs * J + j];
        tot += num / (den + eps);
    }
    if (threadIdx.x < 64 && ch < c) protos[((size_t)b * J + j) * c + ch] = tot / (float)S;
}

// plain masks at feature resolution as pooling weights (MODE 1 of assign_kernel), one thread per pixel
__global__ void mask_assign_kernel(const float* __restrict__ mask, float* __restrict__ A, int n, int h, int w, int H,
                                   int W) {
    const int bs = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int y = i / w, x = i - y * w;
    const int sy = nearest_src(y, H, h), sx = nearest_src(x, W, w);
    const float* mk = mask + (size_t)bs * 2 * H * W;
    A[((size_t)bs * 2 + 0) * n + i] = mk[(size_t)sy * W + sx];
    A[((size_t)bs * 2 + 1) * n + i] = mk[(size_t)H * W + (size_t)sy * W + sx];
}

// -----------------------------------------------------------------------------------------------
// Pixel-row x small-matrix products on the matrix cores.  Both per-pixel passes of the prototype head
// have the same shape: every pixel's c-vector meets a fixed c x 2p matrix -- the L2-normalised
// prototypes for the cosine map (networks/pemp_stage1.py:214-222, 256-260), the MPM centres for the
// soft assignment (pemp_stage1.py:205-207) -- and the feature map is streamed from HBM exactly once.
// One wave owns 16 pixels: D[16 px][16 cols] += X[16 px][4 k] * T[4 k][16 cols] on
// v_mfma_f32_16x16x4_f32, columns 0..2p-1 real (rows of an LDS table, padded so the 16-B reads of the
// 2p+1 rows fall on different banks), the others read an all-zero row.  Lane (r = l&15, q = l>>4)
// reads float4 #q of every 16-float chunk of pixel r's row -- a wave-wide load is 16 rows x 64
// contiguous bytes -- and feeds it to four MFMAs (k = {16t + 4q + e}); |x|^2 comes from the same
// registers.  The kernel is latency-bound unless many loads are in flight: rows are fetched in groups
// of 8 chunks (8 KB per wave), two groups deep, and the first group is issued before the table is built.
typedef __attribute__((ext_vector_type(4))) float hv4f;
constexpr int SQ = 64;                 // channels per stage pass
constexpr int SLD = SQ + 4;            // stage row stride in floats: 16-B reads of 16 rows hit 64 different banks
constexpr int NB = 2;                  // passes in flight (4 KB per wave each)
constexpr int TV = MAXJ * 64 * MAXCL / 256;   // table elements per thread

// 16 pixel rows x c channels through a wave-private LDS stage, SQ channels at a time.  HBM side: lane l reads
// float4 #(l & 15) of row 4u + (l >> 4), u = 0..3 -- a wave-wide load is 4 rows x 256 contiguous bytes, NB
// passes (16 KB per wave) in flight.  MFMA side: lane (r = l&15, q = l>>4) reads float4 #q of every 16-float
// chunk of row r from the stage and the matching table entries, and issues four MFMAs (k = {16t + 4q + e}).
// NQ = c / SQ is a template parameter: with every load and every pass known at compile time the s_waitcnt
// vmcnt bookkeeping is exact (a runtime trip count makes the compiler wait for ALL outstanding loads per pass).
template <int NQ>
struct RowTile {
    const float* base;                 // image base + 4 * (l & 15)
    int off[4];                        // element offset of row 4u + (l >> 4) (clamped to the last pixel)
    hv4f buf[NB][4];
    __device__ __forceinline__ void init(const float* img, int ldf, int i0, int n, int lane) {
        base = img + 4 * (lane & 15);
#pragma unroll
        for (int u = 0; u < 4; ++u) off[u] = min(i0 + 4 * u + (lane >> 4), n - 1) * ldf;
#pragma unroll
        for (int s = 0; s < NB; ++s)
            if (s < NQ) load(s, s);
    }
    __device__ __forceinline__ void load(int s, int qq) {
#pragma unroll
        for (int u = 0; u < 4; ++u) buf[s][u] = *(const hv4f*)(base + off[u] + qq * SQ);
    }
    template <bool NORM>
    __device__ __forceinline__ void pass(int s, int qq, float* stage, const float* bp, int lane, hv4f& acc, float& ss) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *(hv4f*)(stage + (4 * u + (lane >> 4)) * SLD + 4 * (lane & 15)) = buf[s][u];
        if (qq + NB < NQ) load(s, qq + NB);
        __builtin_amdgcn_wave_barrier();
        const float* xr = stage + (lane & 15) * SLD + 4 * (lane >> 4);
#pragma unroll
        for (int t = 0; t < SQ / 16; ++t) {
            const hv4f xv = *(const hv4f*)(xr + 16 * t);
            const hv4f pv = *(const hv4f*)(bp + qq * SQ + 16 * t);
            if (NORM) ss += (xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.x, pv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.y, pv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.z, pv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.w, pv.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    template <bool NORM>
    __device__ __forceinline__ void run(float* stage, const float* bp, int lane, hv4f& acc, float& ss) {  // first NB passes loaded
#pragma unroll
        for (int qq = 0; qq < NQ; qq += NB) {
#pragma unroll
            for (int s = 0; s < NB; ++s)
                if (qq + s < NQ) pass<NORM>(s, qq + s, stage, bp, lane, acc, ss);
        }
    }
};
// D[i][j] of the 16x16 tile lives in lane 16*(i>>2) + j, register i&3: hand row (l & 15) to lane l.
__device__ __forceinline__ void rows_to_lanes(const hv4f& acc, int lane, float (&v)[MAXJ]) {
    const int row = lane & 15, src = 16 * (row >> 2), e = row & 3;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const float t0 = __shfl(acc[0], src + j, 64), t1 = __shfl(acc[1], src + j, 64);
        const float t2 = __shfl(acc[2], src + j, 64), t3 = __shfl(acc[3], src + j, 64);
        v[j] = e == 0 ? t0 : e == 1 ? t1 : e == 2 ? t2 : t3;
    }
}
static inline bool mfma_rows_ok(int c) { return c == 512 || c == 256 || c == 128 || c == 64; }   // instantiated NQ
static inline size_t proj_lds_bytes(int J, int c) { return ((size_t)(J + 1) * (c + 8) + 4 * 16 * SLD) * sizeof(float); }
// squared norms of the table rows, every wave for itself (lane l of the result holds row l & 7)
__device__ __forceinline__ float table_sqnorm(const float* tab, int ldt, int J, int c, int lane) {
    float s[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        s[j] = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCL / 4; ++k) {                        // branch-free: rows >= J are the zero row
            const int ch = 4 * lane + 256 * k;
            const hv4f v = *(const hv4f*)(tab + min(j, J) * ldt + (ch < c ? ch : 0));
            s[j] += ch < c ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
        }
    }
    return wave_sum8(s);
}

// cosine map x dist_scalar, group maxima, response index: the MFMA outer product with fused L2 normalisation.
template <int NQ>
__global__ __launch_bounds__(256, 5) void cosine_mfma_kernel(const float* __restrict__ qry, int ldf,
                                                          const float* __restrict__ protos, float* __restrict__ pred,
                                                          uint8_t* __restrict__ resp, int n, int c, int p,
                                                          float scalar) {
    extern __shared__ __attribute__((aligned(16))) float pnl[];      // [(2p + 1)][c + 8] prototypes, last row zeros | stages
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int J = 2 * p, ldt = c + 8;
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    const int r = lane & 15, q = lane >> 4;
    float* stage = pnl + (J + 1) * ldt + wave * 16 * SLD;
    // the table's loads go first: waiting for them must not wait for the 16 KB of row data behind them
    const float* pb = protos + (size_t)b * J * c;
    float tv[TV];
#pragma unroll
    for (int k = 0; k < TV; ++k) tv[k] = pb[min((int)threadIdx.x + 256 * k, J * c - 1)];
    RowTile<NQ> rt;
    rt.init(qry + (size_t)b * n * ldf, ldf, i0, n, lane);
#pragma unroll
    for (int k = 0; k < TV; ++k) {
        const int t = threadIdx.x + 256 * k;
        if (t < J * c) pnl[(t / c) * ldt + t % c] = tv[k];
    }
    for (int ch = threadIdx.x; ch < c; ch += 256) pnl[J * ldt + ch] = 0.f;
    __syncthreads();
    // cos = x.p / (max(|x|, eps) max(|p|, eps)): the prototype norms are applied in the epilogue
    const float pn2 = table_sqnorm(pnl, ldt, J, c, lane);
    hv4f acc = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    rt.template run<true>(stage, pnl + (size_t)min(r, J) * ldt + 4 * q, lane, acc, ss);
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);                                    // lane l: |x|^2 of pixel (l & 15)
    const float f = scalar / fmaxf(sqrtf(ss), 1e-8f);
    float v[MAXJ];
    rows_to_lanes(acc, lane, v);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) v[j] = v[j] / fmaxf(sqrtf(__shfl(pn2, j, 64)), 1e-8f) * f;
    const int i = i0 + r;
    if (q == 0 && i < n) {
        float best[2] = {0.f, 0.f};              // [0]: fg rows [0,p), [1]: bg rows [p,2p); first maximum wins
        int bi[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)            // static register indices: no scratch for v[]
            if (j < J) {
                const bool fg = j < p;
                const int first = fg ? 0 : p;
                if (fg) {
                    if (j == first || v[j] > best[0]) { best[0] = v[j]; bi[0] = j; }
                } else {
                    if (j == first || v[j] > best[1]) { best[1] = v[j]; bi[1] = j - p; }
                }
            }
        pred[((size_t)b * 2 + 0) * n + i] = best[1];
        pred[((size_t)b * 2 + 1) * n + i] = best[0];
        if (resp) resp[(size_t)b * n + i] = (uint8_t)(best[0] > best[1] ? bi[0] + 3 : bi[1]);
    }
}

// MPM soft assignment on the same row stream (MODE 0 of assign_kernel): softmax_j(-|x - c_j|^2) within
// the fg and the bg group equals softmax_j(2 x.c_j - |c_j|^2) -- |x|^2 is common to a group and drops
// out -- so the MFMA accumulates x.c_j and the epilogue finishes 16 pixels on 16 lanes.
template <int NQ>
__global__ __launch_bounds__(256, 5) void assign_mfma_kernel(const float* __restrict__ feat, int ldf,
                                                          const float* __restrict__ mask,
                                                          const float* __restrict__ ctr, float* __restrict__ A,
                                                          int n, int h, int w, int H, int W, int c, int p) {
    extern __shared__ __attribute__((aligned(16))) float tab[];      // [(2p + 1)][c + 8]: centres, last row zeros | stages
    const int bs = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int J = 2 * p, ldt = c + 8;
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    const int r = lane & 15, q = lane >> 4;
    float* stage = tab + (J + 1) * ldt + wave * 16 * SLD;
    float tv[TV];                                                    // table loads first (see cosine_mfma_kernel)
#pragma unroll
    for (int k = 0; k < TV; ++k) tv[k] = ctr[min((int)threadIdx.x + 256 * k, J * c - 1)];
    RowTile<NQ> rt;
    rt.init(feat + (size_t)bs * n * ldf, ldf, i0, n, lane);
#pragma unroll
    for (int k = 0; k < TV; ++k) {                                   // ctr is [c][2p]
        const int t = threadIdx.x + 256 * k;
        if (t < J * c) tab[(t % J) * ldt + t / J] = tv[k];
    }
    for (int ch = threadIdx.x; ch < c; ch += 256) tab[J * ldt + ch] = 0.f;
    __syncthreads();
    hv4f acc = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    rt.template run<false>(stage, tab + (size_t)min(r, J) * ldt + 4 * q, lane, acc, ss);
    const float cn2 = table_sqnorm(tab, ldt, J, c, lane);            // |c_j|^2, needed by the epilogue only
    float cn[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) cn[j] = __shfl(cn2, j, 64);
    float v[MAXJ];
    rows_to_lanes(acc, lane, v);
    const int i = i0 + r;
    if (q == 0 && i < n) {
        const int y = i / w, x = i - y * w;
        const int sy = nearest_src(y, H, h), sx = nearest_src(x, W, w);
        const float* mk = mask + (size_t)bs * 2 * H * W;
        const float mg[2] = {mk[(size_t)sy * W + sx], mk[(size_t)H * W + (size_t)sy * W + sx]};
        float* out = A + ((size_t)bs * J) * n + i;
        float l[MAXJ], mx[2] = {-INFINITY, -INFINITY}, sm[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)            // static register indices throughout; group 0 = rows [0,p)
            if (j < J) {
                l[j] = 2.f * v[j] - cn[j];
                if (j < p) mx[0] = fmaxf(mx[0], l[j]);
                else mx[1] = fmaxf(mx[1], l[j]);
            }
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
            if (j < J) {
                l[j] = expf(l[j] - (j < p ? mx[0] : mx[1]));
                if (j < p) sm[0] += l[j];
                else sm[1] += l[j];
            }
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
            if (j < J) out[(size_t)j * n] = (l[j] / (j < p ? sm[0] : sm[1])) * (j < p ? mg[0] : mg[1]);
    }
}

// Masked pooling on the matrix cores: part[j][ch] = sum_i A[j][i] x[i][ch] over one chunk of PCHUNK pixels
// as D[16 rows j][16 ch] += A[j][4 px] * X[4 px][16 ch] (v_mfma_f32_16x16x4_f32, rows >= 2p read a zero row
// of the LDS weight table).  Lane (m = l&15, k = l>>4) reads float4 #m of pixel k's 64-channel slab -- a
// wave-wide load is 4 pixels x 256 contiguous bytes -- and feeds four MFMAs with four accumulators (channels
// 4m + e); wave v owns slabs v and v + 4, so a block covers c <= 512 channels and every row is read once.
// Four pixel-quads (8 KB per wave) are fetched per group, two groups in flight.
__global__ __launch_bounds__(256, 5) void pool_mfma_kernel(const float* __restrict__ feat, int ldf,
                                                        const float* __restrict__ A, float* __restrict__ part,
                                                        float* __restrict__ asum, int n, int c, int J, int nchunks) {
    __shared__ float As[MAXJ + 1][PCHUNK];
    constexpr int QG = 2;                                            // pixel quads per prefetch group
    constexpr int NG = PCHUNK / (4 * QG);
    const int bs = blockIdx.y, ck = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, k = lane >> 4;
    const int i0 = ck * PCHUNK;
    const int np = min(PCHUNK, n - i0);
    const int nslab = c / 64;
    const bool s0 = wave < nslab, s1 = wave + 4 < nslab;
    const float* xb = feat + (size_t)bs * n * ldf + 64 * wave + 4 * m;
    hv4f buf[2][QG][2];
    auto load = [&](int s, int g) {
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            const float* xr = xb + (size_t)min(i0 + 4 * (g * QG + u) + k, n - 1) * ldf;
            if (s0) buf[s][u][0] = *(const hv4f*)xr;
            if (s1) buf[s][u][1] = *(const hv4f*)(xr + 256);
        }
    };
    load(0, 0);
    for (int t = threadIdx.x; t < (MAXJ + 1) * PCHUNK; t += 256) {
        const int j = t / PCHUNK, i = t - j * PCHUNK;
        As[j][i] = (j < J && i < np) ? A[((size_t)bs * J + j) * n + i0 + i] : 0.f;
    }
    __syncthreads();
    hv4f acc[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s][e] = (hv4f){0.f, 0.f, 0.f, 0.f};
    const float* ap = &As[min(m, J)][k];
    auto mac = [&](int s, int g) {
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            const float a = ap[4 * (g * QG + u)];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                if (sl == 0 ? s0 : s1) {
                    const hv4f xv = buf[s][u][sl];
                    acc[sl][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.x, acc[sl][0], 0, 0, 0);
                    acc[sl][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.y, acc[sl][1], 0, 0, 0);
                    acc[sl][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.z, acc[sl][2], 0, 0, 0);
                    acc[sl][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv.w, acc[sl][3], 0, 0, 0);
                }
        }
    };
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) load(1, g + 1);
        mac(0, g);
        if (g + 2 < NG) load(0, g + 2);
        if (g + 1 < NG) mac(1, g + 1);
    }
    // D[j][m]: lane (m, k) holds rows j = 4k + e' of channel 4m + e in acc[.][e][e']
    float* pb = part + ((size_t)bs * nchunks + ck) * J * c;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        if (sl == 0 ? s0 : s1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * k + e;
                if (j < J) {
                    const hv4f o = {acc[sl][0][e], acc[sl][1][e], acc[sl][2][e], acc[sl][3][e]};
                    *(hv4f*)(pb + (size_t)j * c + 64 * (wave + 4 * sl) + 4 * m) = o;
                }
            }
    if (threadIdx.x < J) {
        float s = 0.f;
        for (int i = 0; i < np; ++i) s += As[threadIdx.x][i];
        asum[((size_t)bs * nchunks + ck) * J + threadIdx.x] = s;
    }
}

// -----------------------------------------------------------------------------------------------
// cosine map + group max, VALU variant (c not a multiple of 8, or PEMP_HEAD_VALU set): one wave per query pixel.
// torch>=2 F.cosine_similarity: each vector is divided by max(||.||, 1e-8), then dotted.
__global__ __launch_bounds__(256) void cosine_kernel(const float* __restrict__ qry, int ldf,
                                                     const float* __restrict__ protos, float* __restrict__ pred,
                                                     uint8_t* __restrict__ resp, int n, int c, int p, float scalar) {
    __shared__ float pn[MAXJ][64 * MAXCL];
    __shared__ float nrm[MAXJ];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int J = 2 * p;
    const float* pb = protos + (size_t)b * J * c;
    for (int j = wave; j < J; j += 4) {
        float s = 0.f;
        for (int ch = lane; ch < c; ch += 64) {
            float v = pb[(size_t)j * c + ch];
            s += v * v;
        }
        s = wave_sum(s);
        if (lane == 0) nrm[j] = fmaxf(sqrtf(s), 1e-8f);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < J * c; t += 256) {
        int j = t / c, ch = t - j * c;
        pn[j][ch] = pb[t] / nrm[j];
    }
    __syncthreads();
    const int ncl = (c + 255) / 256;
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        const float* xp = qry + ((size_t)b * n + i) * ldf;
        float xv[MAXCL];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            float4 v = (t < ncl && ch < c) ? *(const float4*)(xp + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            xv[t * 4 + 0] = v.x; xv[t * 4 + 1] = v.y; xv[t * 4 + 2] = v.z; xv[t * 4 + 3] = v.w;
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        ss = wave_sum(ss);
        const float nx = fmaxf(sqrtf(ss), 1e-8f);
        float dot[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) dot[j] = 0.f;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            if (t < ncl && ch < c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xn = xv[t * 4 + e] / nx;
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j)
                        if (j < J) dot[j] += xn * pn[j][ch + e];
                }
            }
        }
        {
            const float tot = wave_sum8(dot);
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) dot[j] = __shfl(tot, j, 64) * scalar;
        }
        if (lane == 0) {
            float best[2];
            int bi[2];
            for (int g = 0; g < 2; ++g) {  // g = 0: fg rows [0,p), g = 1: bg rows [p,2p)
                best[g] = dot[g * p];
                bi[g] = 0;
                for (int j = 1; j < p; ++j)
                    if (dot[g * p + j] > best[g]) {
                        best[g] = dot[g * p + j];
                        bi[g] = j;
                    }
            }
            pred[((size_t)b * 2 + 0) * n + i] = best[1];  // channel 0 = bg
            pred[((size_t)b * 2 + 1) * n + i] = best[0];  // channel 1 = fg
            if (resp) resp[(size_t)b * n + i] = (uint8_t)(best[0] > best[1] ? bi[0] + 3 : bi[1]);
        }
    }
}

// masks[b][0][i] = [argmax_ch pred[b][.][i] == 1], masks[b][1][i] = [argmax == 0]  (channel 0 wins ties, as torch.argmax):
// the query masks of PANet's alignment branch (networks/panet.py:169-171)
__global__ void argmax_masks_kernel(const float* __restrict__ pred, float* __restrict__ masks, int B, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n) return;
    const int b = t / n, i = t - b * n;
    const float fg = pred[((size_t)b * 2 + 1) * n + i] > pred[((size_t)b * 2 + 0) * n + i] ? 1.f : 0.f;
    masks[((size_t)b * 2 + 0) * n + i] = fg;
    masks[((size_t)b * 2 + 1) * n + i] = 1.f - fg;
}

// -----------------------------------------------------------------------------------------------
__global__ void upsample_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int BC, int h, int w,
                                         int Ho, int Wo) {
    long long total = (long long)BC * Ho * Wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int X = (int)(i % Wo);
        long long t = i / Wo;
        int Y = (int)(t % Ho);
        int bc = (int)(t / Ho);
        out[i] = bilerp(in + (size_t)bc * h * w, w, bilin(Y, h, Ho), bilin(X, w, Wo));
    }
}

__global__ void upsample_nearest_kernel(const uint8_t* __restrict__ in, int64_t* __restrict__ out, int B, int h,
                                        int w, int Ho, int Wo) {
    long long total = (long long)B * Ho * Wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int X = (int)(i % Wo);
        long long t = i / Wo;
        int Y = (int)(t % Ho);
        int b = (int)(t / Ho);
        out[i] = in[((size_t)b * h + nearest_src(Y, h, Ho)) * w + nearest_src(X, w, Wo)];
    }
}

// eval tail: upsample + argmax + CE partials + tp/fp/fn partials.  part[b][blk][8] doubles.
__global__ __launch_bounds__(256) void eval_tail_kernel(const float* __restrict__ pred,
                                                        const int64_t* __restrict__ target,
                                                        const float* __restrict__ weight,
                                                        uint8_t* __restrict__ pred_out, float* __restrict__ logits,
                                                        double* __restrict__ part, int h, int w, int Ho, int Wo) {
    __shared__ double red[4][8];
    const int b = blockIdx.y;
    const int npix = Ho * Wo;
    const float* p0 = pred + (size_t)b * 2 * h * w;
    const float* p1 = p0 + h * w;
    double ce = 0.0, wsum = 0.0;
    int cnt[7] = {0, 0, 0, 0, 0, 0, 0};  // valid, tp0, fp0, fn0, tp1, fp1, fn1
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        int Y = i / Wo, X = i - Y * Wo;
        Bilin by = bilin(Y, h, Ho), bx = bilin(X, w, Wo);
        float l0 = bilerp(p0, w, by, bx), l1 = bilerp(p1, w, by, bx);
        int am = l1 > l0 ? 1 : 0;
        pred_out[(size_t)b * npix + i] = (uint8_t)am;
        if (logits) {
            logits[((size_t)b * 2 + 0) * npix + i] = l0;
            logits[((size_t)b * 2 + 1) * npix + i] = l1;
        }
        if (target) {
            int t = (int)target[(size_t)b * npix + i];
            // CELossDT (core/losses.py:33-43): per-pixel weights; the denominator sums the weight of EVERY
            // pixel, ignored ones included (their CE term is zero)
            const float wgt = weight ? weight[(size_t)b * npix + i] : 1.f;
            wsum += (double)wgt;
            if (t != 255) {
                float m = fmaxf(l0, l1);
                float lse = m + logf(expf(l0 - m) + expf(l1 - m));
                ce += (double)((lse - (t == 1 ? l1 : l0)) * wgt);
                cnt[0]++;
                for (int j = 0; j < 2; ++j) {
                    cnt[1 + 3 * j] += (am == j && t == j);
                    cnt[2 + 3 * j] += (am == j && t != j);
                    cnt[3 + 3 * j] += (am != j && t == j);
                }
            }
        }
    }
    double v[8];
    v[0] = ce;
    for (int k = 0; k < 7; ++k) v[k + 1] = (double)cnt[k];
    if (weight) v[1] = wsum;      // loss denominator: sum of weights instead of the valid-pixel count
    for (int k = 0; k < 8; ++k) {
        double x = v[k];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        int k = threadIdx.x;
        part[((size_t)b * gridDim.x + blockIdx.x) * 8 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}
// one wave per (episode, statistic): lane-strided partial sums then a fixed butterfly -> deterministic
__global__ void eval_tail_final_kernel(const double* __restrict__ part, double* __restrict__ stats, int nblk) {
    const int b = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
    double s = 0.0;
    for (int i = lane; i < nblk; i += 64) s += part[((size_t)b * nblk + i) * 8 + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) stats[b * 8 + k] = s;
}

static inline int tail_blocks(int Ho, int Wo) {
    int nb = cdiv(Ho * Wo, 256 * 4);
    return nb < 1 ? 1 : (nb > 256 ? 256 : nb);
}

}  // namespace pemp

using namespace pemp;

// workspace layout for pooling: A[BS][J][n] | part[BS][nchunks][J][c] | asum[BS][nchunks][J] | msum[BS][2]
static size_t pool_ws_bytes(int BS, int n, int c, int J) { return pool_ws_floats(BS, n, c, J) * sizeof(float); }

extern "C" size_t pemp_mpm_workspace_bytes(int B, int S, int n, int c, int p) {
    return pool_ws_bytes(B * S, n, c, 2 * p);
}
extern "C" size_t pemp_map_workspace_bytes(int B, int S, int n, int c) { return pool_ws_bytes(B * S, n, c, 2); }

static int pooled_protos(int mode, const float* feat, int ldf, const float* mask, const float* ctr, float* protos,
                         void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W, int c, int p,
                         hipStream_t st) {
    const int BS = B * S, n = h * w;
    const int J = mode == 0 ? 2 * p : 2;
    PEMP_REQUIRE(feat && mask && protos && ws, "protos: null pointer");
    PEMP_REQUIRE(B > 0 && S > 0 && h > 0 && w > 0 && H > 0 && W > 0, "protos: bad dims");
    PEMP_REQUIRE(c > 0 && c % 4 == 0 && c <= 64 * MAXCL && ldf >= c && ldf % 4 == 0, "protos: c=%d must be a multiple of 4 and <= %d", c, 64 * MAXCL);
    PEMP_REQUIRE(J >= 2 && J <= MAXJ, "protos: 2p=%d not in 2..%d", J, MAXJ);
    PEMP_REQUIRE(ws_bytes >= pool_ws_bytes(BS, n, c, J), "protos: workspace too small");
    PEMP_REQUIRE(((uintptr_t)feat & 15) == 0, "protos: feat must be 16-byte aligned");
    const int nck = nchunks_of(n);
    const PoolWs L = pool_ws_layout(ws, BS, n, c, J);
    static const bool force_valu = getenv("PEMP_HEAD_VALU") != nullptr;       // A/B switch for measurements
    float *A = L.A, *part = L.part, *asum = L.asum, *msum = L.msum;
    // few, long-lived blocks: every block first loads its lanes' slice of ctr (48 values per lane)
    const int ablk = min(cdiv(n, 4), max(1, 512 / BS));
    if (mode == 0) {
        PEMP_REQUIRE(ctr, "protos: ctr is null");
        if (mfma_rows_ok(c) && !force_valu) {
            const dim3 grid(cdiv(cdiv(n, 16), 4), BS);
#define PEMP_ASG(NQ)                                                                                                  \
    hipLaunchKernelGGL(assign_mfma_kernel<NQ>, grid, dim3(256), proj_lds_bytes(J, c), st, feat, ldf, mask, ctr, A, n, h, w, H, \
                       W, c, p)
            if (c == 512) PEMP_ASG(8);
            else if (c == 256) PEMP_ASG(4);
            else if (c == 128) PEMP_ASG(2);
            else PEMP_ASG(1);
#undef PEMP_ASG
        }
        else
            hipLaunchKernelGGL(assign_kernel<0>, dim3(ablk, BS), dim3(256), 0, st, feat, ldf, mask, ctr, A, n, h, w, H, W, c, p);
    } else if (mode == 1) {
        hipLaunchKernelGGL(mask_assign_kernel, dim3(cdiv(n, 256), BS), dim3(256), 0, st, mask, A, n, h, w, H, W);
    } else {
        hipLaunchKernelGGL(adjoint_mask_kernel, dim3(cdiv(2 * n, 256), BS), dim3(256), 0, st, mask, A, n, h, w, H, W);
        hipLaunchKernelGGL(mask_sum_kernel, dim3(BS * 2), dim3(1024), 0, st, mask, msum, H * W);
    }
    int e = launch_status("protos/assign");
    if (e) return e;
    if (c % 64 == 0 && !force_valu)
        hipLaunchKernelGGL(pool_mfma_kernel, dim3(nck, BS), dim3(256), 0, st, feat, ldf, A, part, asum, n, c, J, nck);
    else
        hipLaunchKernelGGL(pool_partial_kernel, dim3(nck, BS), dim3(256), 0, st, feat, ldf, A, part, asum, n, c, J, nck);
    hipLaunchKernelGGL(pool_final_kernel, dim3(J, B, cdiv(c, 64)), dim3(256), 0, st, part, asum, mode == 2 ? msum : (const float*)nullptr,
                       protos, S, c, J, nck, mode == 0 ? 1e-6f : 1e-5f);
    return launch_status("protos/pool");
}

extern "C" int pemp_mpm_protos_f32(const float* feat, int ldf, const float* mask, const float* ctr, float* protos,
                                   void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W, int c, int p,
                                   void* stream) {
    PEMP_REQUIRE(p >= 1 && p <= MAXJ / 2, "mpm_protos: p=%d not in 1..%d", p, MAXJ / 2);
    return pooled_protos(0, feat, ldf, mask, ctr, protos, ws, ws_bytes, B, S, h, w, H, W, c, p, (hipStream_t)stream);
}

extern "C" int pemp_masked_avg_pool_f32(const float* feat, int ldf, const float* mask, float* protos, void* ws,
                                        size_t ws_bytes, int B, int S, int h, int w, int H, int W, int c, int full_res,
                                        void* stream) {
    return pooled_protos(full_res ? 2 : 1, feat, ldf, mask, nullptr, protos, ws, ws_bytes, B, S, h, w, H, W, c, 1,
                         (hipStream_t)stream);
}

extern "C" int pemp_cosine_proto_max_f32(const float* qry, int ldf, const float* protos, float* pred, uint8_t* resp,
                                         int B, int n, int c, int p, float dist_scalar, void* stream) {
    PEMP_REQUIRE(qry && protos && pred, "cosine: null pointer");
    PEMP_REQUIRE(B > 0 && n > 0 && p >= 1 && 2 * p <= MAXJ, "cosine: bad dims");
    PEMP_REQUIRE(c > 0 && c % 4 == 0 && c <= 64 * MAXCL && ldf >= c && ldf % 4 == 0, "cosine: c=%d must be a multiple of 4 and <= %d", c, 64 * MAXCL);
    PEMP_REQUIRE(((uintptr_t)qry & 15) == 0, "cosine: qry must be 16-byte aligned");
    static const bool force_valu = getenv("PEMP_HEAD_VALU") != nullptr;     // A/B switch for measurements
    if (mfma_rows_ok(c) && !force_valu) {
        // MFMA outer product: 16 query pixels x (2p prototypes padded to 16 columns) per wave
        const size_t lds = proj_lds_bytes(2 * p, c);
        const dim3 grid(cdiv(cdiv(n, 16), 4), B);
#define PEMP_COS(NQ)                                                                                                  \
    hipLaunchKernelGGL(cosine_mfma_kernel<NQ>, grid, dim3(256), lds, (hipStream_t)stream, qry, ldf, protos, pred, resp, n, c, \
                       p, dist_scalar)
        if (c == 512) PEMP_COS(8);
        else if (c == 256) PEMP_COS(4);
        else if (c == 128) PEMP_COS(2);
        else PEMP_COS(1);
#undef PEMP_COS
        return launch_status("cosine_mfma");
    }
    hipLaunchKernelGGL(cosine_kernel, dim3(min(cdiv(n, 4), max(1, 4096 / B)), B), dim3(256), 0, (hipStream_t)stream, qry, ldf, protos,
                       pred, resp, n, c, p, dist_scalar);
    return launch_status("cosine");
}

extern "C" int pemp_argmax_masks_f32(const float* pred, float* masks, int B, int n, void* stream) {
    PEMP_REQUIRE(pred && masks && B > 0 && n > 0, "argmax_masks: bad arguments");
    hipLaunchKernelGGL(argmax_masks_kernel, dim3(cdiv(B * n, 256)), dim3(256), 0, (hipStream_t)stream, pred, masks, B, n);
    return launch_status("argmax_masks");
}

extern "C" int pemp_upsample_bilinear_ac_f32(const float* pred, float* out, int B, int C, int h, int w, int Ho, int Wo,
                                             void* stream) {
    PEMP_REQUIRE(pred && out && B > 0 && C > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0, "upsample: bad arguments");
    long long total = (long long)B * C * Ho * Wo;
    int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, out, B * C, h, w, Ho, Wo);
    return launch_status("upsample_bilinear");
}

extern "C" int pemp_upsample_nearest_u8_i64(const uint8_t* resp, int64_t* out, int B, int h, int w, int Ho, int Wo,
                                            void* stream) {
    PEMP_REQUIRE(resp && out && B > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0, "upsample_nearest: bad arguments");
    long long total = (long long)B * Ho * Wo;
    int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, resp, out, B, h, w, Ho, Wo);
    return launch_status("upsample_nearest");
}

extern "C" size_t pemp_eval_tail_workspace_bytes(int B, int Ho, int Wo) {
    return (size_t)B * tail_blocks(Ho, Wo) * 8 * sizeof(double);
}

extern "C" int pemp_eval_tail_weighted_f32(const float* pred, const int64_t* target, const float* weight,
                                           uint8_t* pred_out, float* logits_out, double* stats, void* ws,
                                           size_t ws_bytes, int B, int h, int w, int Ho, int Wo, void* stream);

extern "C" int pemp_eval_tail_f32(const float* pred, const int64_t* target, uint8_t* pred_out, float* logits_out,
                                  double* stats, void* ws, size_t ws_bytes, int B, int h, int w, int Ho, int Wo,
                                  void* stream) {
    return pemp_eval_tail_weighted_f32(pred, target, nullptr, pred_out, logits_out, stats, ws, ws_bytes, B, h, w, Ho, Wo,
                                       stream);
}

extern "C" int pemp_eval_tail_weighted_f32(const float* pred, const int64_t* target, const float* weight,
                                           uint8_t* pred_out, float* logits_out, double* stats, void* ws,
                                           size_t ws_bytes, int B, int h, int w, int Ho, int Wo, void* stream) {
    PEMP_REQUIRE(pred && pred_out && stats && ws, "eval_tail: null pointer");
    PEMP_REQUIRE(!weight || target, "eval_tail: weight without target");
    PEMP_REQUIRE(B > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0, "eval_tail: bad dims");
    PEMP_REQUIRE(ws_bytes >= pemp_eval_tail_workspace_bytes(B, Ho, Wo), "eval_tail: workspace too small");
    const int nb = tail_blocks(Ho, Wo);
    hipLaunchKernelGGL(eval_tail_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, pred, target, weight, pred_out,
                       logits_out, (double*)ws, h, w, Ho, Wo);
    hipLaunchKernelGGL(eval_tail_final_kernel, dim3(B, 8), dim3(64), 0, (hipStream_t)stream, (const double*)ws, stats, nb);
    return launch_status("eval_tail");
}
