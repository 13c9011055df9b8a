// Training-path streaming kernels (HBM-bound): train-mode BatchNorm forward/backward with batch
// statistics, ReLU / bias backward, max-pool backward, strided scatter for stride-2 dgrad, and the
// fused clip-norm + SGD-momentum update on flat parameter/gradient buffers.
//
// All reductions are two-stage with a fixed order (per-chunk partials in double, then a serial
// sum over chunks), so every result is deterministic and independent of launch geometry.
#include "common.h"
#include <stdlib.h>

namespace pemp {

constexpr int RCHUNK = 64;    // rows per partial in the per-channel reductions (4 per thread, all loads issued before the sums:
                              // M = 20.8k rows x 256 channels gives 1300 blocks instead of 328, 12 loads in flight per thread)

// -----------------------------------------------------------------------------------------------
// per-channel sums over rows:  out[chunk][0][c] = sum_r a(r,c),  out[chunk][1][c] = sum_r b(r,c)
//   MODE 0 (BN stats):      a = z,            b = z*z
//   MODE 1 (BN backward):   a = g,            b = g * xhat      g = dy * (y > 0 if relu)
//   MODE 2 (bias backward): a = g,            b unused
// block = 16 float4-lanes (64 channels) x 16 row-lanes.
template <int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ p0, int ld0,
                                                     const float* __restrict__ p1, int ld1,
                                                     const float* __restrict__ p2, int ld2,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, double* __restrict__ part,
                                                     int M, int C, int relu, const uint32_t* __restrict__ mask = nullptr) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cl * 4;
    const int r0 = blockIdx.x * RCHUNK;
    const int r1 = min(r0 + RCHUNK, M);
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mu[e] = mean[c + e];
                is[e] = invstd[c + e];
            }
        }
        constexpr int RPT = RCHUNK / 16;
        float4 A[RPT], Y[RPT], Z[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {                 // every load of the thread first ...
            const int r = r0 + rl + 16 * k;
            A[k] = Y[k] = Z[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < r1) {
                A[k] = *(const float4*)(p0 + (size_t)r * ld0 + c);
                if (MODE == 1 && relu && mask) {        // the sign bits of y stand in for y (1/32 of the bytes)
                    const uint32_t b = mask[(size_t)r * (C >> 5) + (c >> 5)] >> (c & 31);
                    Y[k] = make_float4((b & 1u) ? 1.f : 0.f, (b & 2u) ? 1.f : 0.f, (b & 4u) ? 1.f : 0.f, (b & 8u) ? 1.f : 0.f);
                } else
                if (MODE != 0 && relu) Y[k] = *(const float4*)(p1 + (size_t)r * ld1 + c);
                if (MODE == 1) Z[k] = *(const float4*)(p2 + (size_t)r * ld2 + c);
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {                 // ... then the sums, rows in ascending order
            if (r0 + rl + 16 * k >= r1) continue;
            float av[4] = {A[k].x, A[k].y, A[k].z, A[k].w};
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sa[e] += av[e];
                    sb[e] += av[e] * av[e];
                }
            } else {
                if (relu) {
                    const float yv[4] = {Y[k].x, Y[k].y, Y[k].z, Y[k].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = yv[e] > 0.f ? av[e] : 0.f;
                }
                if (MODE == 1) {
                    const float zv[4] = {Z[k].x, Z[k].y, Z[k].z, Z[k].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sa[e] += av[e];
                        sb[e] += av[e] * ((zv[e] - mu[e]) * is[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sa[e] += av[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][rl][cl * 4 + e] = (double)sa[e];
        red[1][rl][cl * 4 + e] = (double)sb[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int k = threadIdx.x >> 6, ch = threadIdx.x & 63;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[k][i][ch];
        if (blockIdx.y * 64 + ch < C) part[((size_t)blockIdx.x * 2 + k) * C + blockIdx.y * 64 + ch] = s;
    }
}

// stage 2 of the column sums: one wave per channel; lane l adds chunks l, l+64, ... in order, then a
// fixed butterfly -> deterministic.  Returns the two totals in every lane.
__device__ __forceinline__ void colsum_total(const double* __restrict__ part, int nchunk, int C, int c, double& s,
                                             double& ss) {
    const int lane = threadIdx.x & 63;
    s = 0.0;
    ss = 0.0;
    for (int k = lane; k < nchunk; k += 64) {
        s += part[((size_t)k * 2 + 0) * C + c];
        ss += part[((size_t)k * 2 + 1) * C + c];
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        ss += __shfl_xor(ss, o, 64);
    }
}

// BN statistics: mean, invstd, running-stat update (momentum form of nn.BatchNorm2d)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ part, int nchunk, int M, int C,
                                                             float eps, float momentum, float* __restrict__ mean,
                                                             float* __restrict__ invstd, float* __restrict__ run_mean,
                                                             float* __restrict__ run_var) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s, ss;
    colsum_total(part, nchunk, C, c, s, ss);
    if ((threadIdx.x & 63) != 0) return;
    const double mu = s / M;
    double var = ss / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unbiased);
    }
}

// Totals of the per-row-tile partial sums a conv epilogue left ([n32][2][C] floats; conv_common.h): a block owns 32 channels,
// thread (rl = tid / 8, q = tid % 8) adds the partial rows rl, rl+32, ... of channel quad q in ascending order (double), the
// 32 row-lanes are then added in ascending order: fixed order, independent of launch geometry.  Returns, in threads 0..31,
// the two totals of channel blockIdx.x*32 + tid.
__device__ __forceinline__ bool partials_total(const float* __restrict__ part, int n32, int C, double& tot0, double& tot1) {
    __shared__ double red[32][65];
    const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 32 + q * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    int k = rl;
    for (; k + 96 < n32; k += 128) {                     // four rows in flight per thread
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *(const float4*)(part + ((size_t)(k + 32 * u) * 2 + 0) * C + c);
            b[u] = *(const float4*)(part + ((size_t)(k + 32 * u) * 2 + 1) * C + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[0] += (double)a[u].x; s[1] += (double)a[u].y; s[2] += (double)a[u].z; s[3] += (double)a[u].w;
            ss[0] += (double)b[u].x; ss[1] += (double)b[u].y; ss[2] += (double)b[u].z; ss[3] += (double)b[u].w;
        }
    }
    for (; k < n32; k += 32) {
        const float4 a = *(const float4*)(part + ((size_t)k * 2 + 0) * C + c);
        const float4 b = *(const float4*)(part + ((size_t)k * 2 + 1) * C + c);
        s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
        ss[0] += (double)b.x; ss[1] += (double)b.y; ss[2] += (double)b.z; ss[3] += (double)b.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[rl][q * 4 + e] = s[e];
        red[rl][32 + q * 4 + e] = ss[e];
    }
    __syncthreads();
    if (threadIdx.x >= 64) return false;
    double t = 0.0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
    tot0 = t;
    tot1 = __shfl(t, (threadIdx.x & 31) + 32, 64);       // lane c < 32 also gets the second total of its channel
    return threadIdx.x < 32;
}

__global__ __launch_bounds__(256) void bn_stats_partials_kernel(const float* __restrict__ part, int n32, int M, int C, float eps,
                                                                float momentum, float* __restrict__ mean,
                                                                float* __restrict__ invstd, float* __restrict__ run_mean,
                                                                float* __restrict__ run_var) {
    double s, ss;
    if (!partials_total(part, n32, C, s, ss)) return;
    const int c = blockIdx.x * 32 + threadIdx.x;
    const double mu = s / M;
    double var = ss / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unbiased);
    }
}

// BatchNorm backward sums from the partials of pemp_conv2d_bnbwd_nhwc_f32: out0 = sum g, out1 = sum g * xhat
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, int n32, int C,
                                                              float* __restrict__ out0, float* __restrict__ out1) {
    double s, ss;
    if (!partials_total(part, n32, C, s, ss)) return;
    const int c = blockIdx.x * 32 + threadIdx.x;
    out0[c] = (float)s;
    out1[c] = (float)ss;
}

// backward sums: out0[c] = sum a, out1[c] = sum b  (fp32 results)
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, int nchunk, int C,
                                                           float* __restrict__ out0, float* __restrict__ out1) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s, ss;
    colsum_total(part, nchunk, C, c, s, ss);
    if ((threadIdx.x & 63) != 0) return;
    if (out0) out0[c] = (float)s;
    if (out1) out1[c] = (float)ss;
}

// y = relu?( (z - mean) * invstd * gamma + beta (+ residual) )   -- ATen's alpha/beta form
__global__ void bn_apply_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ res, int ldr,
                                float* __restrict__ y, int ldy, long long M, int C4, int relu) {
    const long long total = M * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long m = i / C4;
        const float4 v = *(const float4*)(z + m * ldz + c);
        float zv[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float alpha = invstd[c + e] * gamma[c + e];
            const float bt = beta[c + e] - mean[c + e] * alpha;
            o[e] = zv[e] * alpha + bt;
        }
        if (res) {
            const float4 rv = *(const float4*)(res + m * ldr + c);
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        *(float4*)(y + m * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dz = gamma*invstd * (g - sum_g/M - xhat * sum_gx/M), g = dy * (y>0 if relu); optionally also writes g
// (the gradient flowing into the residual branch).
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                    const float* __restrict__ z, int ldz, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
                                    float* __restrict__ dz, int lddz, float* __restrict__ gout, int ldg, long long M,
                                    int C4, int relu) {
    const long long total = M * C4;
    const float invM = 1.f / (float)M;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long m = i / C4;
        const float4 d4 = *(const float4*)(dy + m * lddy + c);
        float g[4] = {d4.x, d4.y, d4.z, d4.w};
        if (relu) {
            const float4 y4 = *(const float4*)(y + m * ldy + c);
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
        }
        const float4 z4 = *(const float4*)(z + m * ldz + c);
        const float zv[4] = {z4.x, z4.y, z4.z, z4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float is = invstd[c + e];
            const float xh = (zv[e] - mean[c + e]) * is;
            o[e] = (g[e] - sum_g[c + e] * invM - xh * (sum_gx[c + e] * invM)) * (is * gamma[c + e]);
        }
        *(float4*)(dz + m * lddz + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (gout) *(float4*)(gout + m * ldg + c) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

// g = dy * (y > 0) (+ add)   -- ReLU backward, optionally accumulating a second gradient stream
__global__ void relu_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                const float* __restrict__ add, int lda, float* __restrict__ g, int ldg, long long M,
                                int C4, int relu) {
    const long long total = M * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long m = i / C4;
        float4 d = *(const float4*)(dy + m * lddy + c);
        if (add) {
            const float4 a = *(const float4*)(add + m * lda + c);
            d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        }
        if (relu) {
            const float4 yv = *(const float4*)(y + m * ldy + c);
            d.x = yv.x > 0.f ? d.x : 0.f;
            d.y = yv.y > 0.f ? d.y : 0.f;
            d.z = yv.z > 0.f ? d.z : 0.f;
            d.w = yv.w > 0.f ? d.w : 0.f;
        }
        *(float4*)(g + m * ldg + c) = d;
    }
}


// -----------------------------------------------------------------------------------------------
// Row-walking forms of the three element-wise passes (C/4 divides 256, i.e. C in {64, 128, 256, 512, 1024}): a thread
// owns ONE channel quad for its whole life -- the per-channel constants sit in registers, there is no 64-bit
// index division per element -- and handles ROWS_PT rows with all loads issued before the arithmetic.  The arithmetic
// per element is the expression of the grid-stride kernels above, so the results are bit-identical.
constexpr int ROWS_PT = 4;

__global__ __launch_bounds__(256) void bn_apply_rows_kernel(const float* __restrict__ z, int ldz,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ res, int ldr, float* __restrict__ y,
                                                            int ldy, int M, int C4, int relu, uint32_t* __restrict__ mask,
                                                            const float* __restrict__ rowmask = nullptr,
                                                            const int* __restrict__ rowcnt = nullptr) {
    const int c = (threadIdx.x % C4) * 4, rpb = 256 / C4;
    const float db_sum = rowmask ? (float)*rowcnt : 1.f, db_numel = (float)M;
    float alpha[4], bt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        alpha[e] = invstd[c + e] * gamma[c + e];
        bt[e] = beta[c + e] - mean[c + e] * alpha[e];
    }
    const int r0 = blockIdx.x * rpb * ROWS_PT + threadIdx.x / C4;
    float4 v[ROWS_PT], rv[ROWS_PT];
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int m = r0 + k * rpb;
        if (m < M) {
            v[k] = *(const float4*)(z + (size_t)m * ldz + c);
            if (res) rv[k] = *(const float4*)(res + (size_t)m * ldr + c);
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int m = r0 + k * rpb;
        uint32_t nib = 0;
        if (m < M) {
            const float zv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = zv[e] * alpha[e] + bt[e];
            if (res) {
                o[0] += rv[k].x; o[1] += rv[k].y; o[2] += rv[k].z; o[3] += rv[k].w;
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            if (rowmask) {          // DropBlock2D behind the BatchNorm: ((y * mask) * numel) / sum(mask), the layer's order and rounding
                const float k = rowmask[m];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __fdiv_rn(__fmul_rn(__fmul_rn(o[e], k), db_numel), db_sum);
            }
            *(float4*)(y + (size_t)m * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) nib |= (o[e] > 0.f ? 1u : 0u) << e;
        }
        if (mask) {
            // bit c % 32 of mask[m][c / 32] = (y[m][c] > 0): the 8 lanes of a row that share a 32-channel word are
            // consecutive (C4 % 8 == 0), one of them stores the word
            uint32_t wv = nib << (4 * (threadIdx.x & 7));
            wv |= __shfl_xor(wv, 1, 64);
            wv |= __shfl_xor(wv, 2, 64);
            wv |= __shfl_xor(wv, 4, 64);
            if ((threadIdx.x & 7) == 0 && m < M) mask[(size_t)m * (C4 >> 3) + (c >> 5)] = wv;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y,
                                                                int ldy, const float* __restrict__ z, int ldz,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ sum_g,
                                                                const float* __restrict__ sum_gx, float* __restrict__ dz, int lddz,
                                                                float* __restrict__ gout, int ldg, int M, int C4, int relu,
                                                                const uint32_t* __restrict__ mask = nullptr) {
    const int c = (threadIdx.x % C4) * 4, rpb = 256 / C4;
    const float invM = 1.f / (float)M;
    float is[4], mu[4], sg[4], sgx[4], isg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        is[e] = invstd[c + e];
        mu[e] = mean[c + e];
        sg[e] = sum_g[c + e] * invM;
        sgx[e] = sum_gx[c + e] * invM;
        isg[e] = is[e] * gamma[c + e];
    }
    const int r0 = blockIdx.x * rpb * ROWS_PT + threadIdx.x / C4;
    float4 d4[ROWS_PT], y4[ROWS_PT], z4[ROWS_PT];
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int m = r0 + k * rpb;
        if (m < M) {
            d4[k] = *(const float4*)(dy + (size_t)m * lddy + c);
            if (relu && mask) {
                const uint32_t b = mask[(size_t)m * (C4 >> 3) + (c >> 5)] >> (c & 31);
                y4[k] = make_float4((b & 1u) ? 1.f : 0.f, (b & 2u) ? 1.f : 0.f, (b & 4u) ? 1.f : 0.f, (b & 8u) ? 1.f : 0.f);
            } else if (relu) y4[k] = *(const float4*)(y + (size_t)m * ldy + c);
            z4[k] = *(const float4*)(z + (size_t)m * ldz + c);
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int m = r0 + k * rpb;
        if (m >= M) continue;
        float g[4] = {d4[k].x, d4[k].y, d4[k].z, d4[k].w};
        if (relu) {
            const float yv[4] = {y4[k].x, y4[k].y, y4[k].z, y4[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
        }
        const float zv[4] = {z4[k].x, z4[k].y, z4[k].z, z4[k].w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (zv[e] - mu[e]) * is[e];
            o[e] = (g[e] - sg[e] - xh * sgx[e]) * isg[e];
        }
        *(float4*)(dz + (size_t)m * lddz + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (gout) *(float4*)(gout + (size_t)m * ldg + c) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

__global__ __launch_bounds__(256) void relu_bwd_rows_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y,
                                                            int ldy, const float* __restrict__ add, int lda,
                                                            float* __restrict__ g, int ldg, int M, int C4, int relu) {
    const int c = (threadIdx.x % C4) * 4, rpb = 256 / C4;
    const int r0 = blockIdx.x * rpb * ROWS_PT + threadIdx.x / C4;
    float4 d[ROWS_PT], a4[ROWS_PT], yv[ROWS_PT];
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int m = r0 + k * rpb;
        if (m < M) {
            d[k] = *(const float4*)(dy + (size_t)m * lddy + c);
            if (add) a4[k] = *(const float4*)(add + (size_t)m * lda + c);
            if (relu) yv[k] = *(const float4*)(y + (size_t)m * ldy + c);
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int m = r0 + k * rpb;
        if (m >= M) continue;
        float4 v = d[k];
        if (add) {
            v.x += a4[k].x; v.y += a4[k].y; v.z += a4[k].z; v.w += a4[k].w;
        }
        if (relu) {
            v.x = yv[k].x > 0.f ? v.x : 0.f;
            v.y = yv[k].y > 0.f ? v.y : 0.f;
            v.z = yv[k].z > 0.f ? v.z : 0.f;
            v.w = yv[k].w > 0.f ? v.w : 0.f;
        }
        *(float4*)(g + (size_t)m * ldg + c) = v;
    }
}

static inline bool rows_form(int C) {
    static const bool off = getenv("PEMP_BN_GRIDSTRIDE") != nullptr;      // A/B switch: the grid-stride kernels
    return !off && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0;
}
static inline int rows_grid(int M, int C) { return cdiv(M, (256 / (C / 4)) * ROWS_PT); }

// max-pool backward as a gather: dx[p] = sum over the output windows that contain p and whose
// (first, scan-order) maximum is p.
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                   int N, int H, int W, int C4, int Ho, int Wo, int k, int s, int p) {
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long long t = i / C4;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const int C = C4 * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // output windows (ho, wo) with ho*s - p <= h < ho*s - p + k
        const int ho_lo = max(0, (h + p - k + s) / s), ho_hi = min(Ho - 1, (h + p) / s);
        const int wo_lo = max(0, (w + p - k + s) / s), wo_hi = min(Wo - 1, (w + p) / s);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int h0 = max(ho * s - p, 0), h1 = min(ho * s - p + k, H);
                const int w0 = max(wo * s - p, 0), w1 = min(wo * s - p + k, W);
                const float4 g4 = *(const float4*)(dy + (((long long)n * Ho + ho) * Wo + wo) * C + c);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
                // first (scan-order) maximum of this window per channel, one 16-byte load per window element
                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bpos[4] = {-1, -1, -1, -1};
                for (int hh = h0; hh < h1; ++hh)
                    for (int ww = w0; ww < w1; ++ww) {
                        const float4 v4 = *(const float4*)(x + (((long long)n * H + hh) * W + ww) * C + c);
                        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (v[e] > best[e] || bpos[e] < 0) {
                                best[e] = v[e];
                                bpos[e] = hh * W + ww;
                            }
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (bpos[e] == h * W + w) acc[e] += gv[e];
            }
        *(float4*)(dx + (((long long)n * H + h) * W + w) * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// dst[n, hs*s, ws*s, :] = src[n, hs, ws, :], zero elsewhere (dgrad of a stride-s 1x1 convolution)
__global__ void scatter_strided_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W,
                                       int Hs, int Ws, int C4, int s) {
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h % s == 0 && w % s == 0 && h / s < Hs && w / s < Ws)
            v = ((const float4*)src)[(((long long)n * Hs + h / s) * Ws + w / s) * C4 + c4];
        ((float4*)dst)[i] = v;
    }
}

// Input-gradient weights of every conv of a flat parameter buffer in one launch: for layer l (KRSC weight W[co][t][ci] at
// params + off) the mirror holds D[ci][T-1-t][co] at mirror + off -- the KRSC weight of the conv that computes the input
// gradient (taps flipped, channels transposed).  One 32 x 32 (co x ci) tile of one tap per block, transposed through LDS:
// reads run along ci, writes along co, both in 128-byte rows.  table[l] = {off, cout, taps, cin, first tile, tiles along ci}.
__global__ __launch_bounds__(256) void dgrad_mirror_kernel(const float* __restrict__ params, float* __restrict__ mirror,
                                                           const int* __restrict__ table, int L) {
    __shared__ float tile[32][33];
    const int b = blockIdx.x;
    int lo = 0, hi = L - 1;                     // last layer whose first tile is <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 6 + 4] <= b) lo = mid; else hi = mid - 1;
    }
    const int* e = table + lo * 6;
    const int off = e[0], cout = e[1], taps = e[2], cin = e[3], tci = e[5];
    int t = b - e[4];
    const int ti = t % tci;
    t /= tci;
    const int tco = (cout + 31) >> 5;
    const int to = t % tco, tap = t / tco;
    const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
    const float* src = params + off;
    float* dst = mirror + off;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = to * 32 + y0 + 8 * k, ci = ti * 32 + x;
        tile[y0 + 8 * k][x] = (co < cout && ci < cin) ? src[((size_t)co * taps + tap) * cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = ti * 32 + y0 + 8 * k, co = to * 32 + x;
        if (ci < cin && co < cout) dst[((size_t)ci * taps + (taps - 1 - tap)) * cout + co] = tile[x][y0 + 8 * k];
    }
}

// y[n][i][c] = v[n][c] / HW  (backward of the global average pool) added into dst
__global__ void gap_bwd_add_kernel(const float* __restrict__ v, float* __restrict__ dst, int ld, int HW, int C4,
                                   long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long m = i / C4;
        const long long n = m / HW;
        const float4 g = ((const float4*)v)[n * C4 + c4];
        float4* d = (float4*)(dst + m * ld + c4 * 4);
        float4 o = *d;
        const float inv = 1.f / (float)HW;
        o.x += g.x * inv; o.y += g.y * inv; o.z += g.z * inv; o.w += g.w * inv;
        *d = o;
    }
}

// ---- optimizer ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ g, long long n,
                                                            double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long b0 = blockIdx.x * per, b1 = min(b0 + per, n);
    float acc = 0.f;
    int cnt = 0;
    for (long long i = b0 + threadIdx.x; i < b1; i += 256) {
        const float v = g[i];
        acc += v * v;
        if (++cnt == 64) {
            s += (double)acc;
            acc = 0.f;
            cnt = 0;
        }
    }
    s += (double)acc;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// one wave: lane-strided partial sums, then a fixed butterfly (deterministic)
__global__ void sqsum_final_kernel(const double* __restrict__ part, int nb, float* __restrict__ norm_out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) norm_out[0] = (float)sqrt(s);
}
// torch.nn.utils.clip_grad_norm_ + torch.optim.SGD (momentum, weight decay, no nesterov) in one pass
__global__ void sgd_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                long long n, const float* __restrict__ norm, float max_norm, float lr, float mom,
                                float wd, int first_step, float grad_scale, int nesterov) {
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm[0] * grad_scale + 1e-6f), 1.f);
    coef *= grad_scale;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float w = p[i];
        float d = g[i] * coef + wd * w;
        const float b = first_step ? d : mom * buf[i] + d;
        buf[i] = b;
        p[i] = w - lr * (nesterov ? d + mom * b : b);
    }
}

// torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (L2 weight decay added to the gradient, no amsgrad) in one pass, in the
// operation order of ATen's Adam (torch/optim/adam.py, _single_tensor_adam / _multi_tensor_adam -- the same arithmetic):
//   g = coef * g + wd * p;  m.lerp_(g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= step_size * m / (sqrt(v) / sqrt(bias2) + eps)
// with step_size = lr / bias1.  The hyper-parameters reach ATen as Python floats (doubles) and are narrowed to float once
// per scalar -- 1 - beta in double FIRST: float(1 - 0.999) and 1.f - 0.999f differ by 1.3e-5 relative -- so the host does
// exactly that (omb1 / omb2 below) instead of subtracting in float.
__global__ void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long long n, const float* __restrict__ norm, float max_norm,
                                 float step_size, float bias2_sqrt, float omb1, float b2, float omb2, float eps, float wd,
                                 float grad_scale) {
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm[0] * grad_scale + 1e-6f), 1.f);
    coef *= grad_scale;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float w = p[i];
        const float d = __fmaf_rn(wd, w, g[i] * coef);                      // grad.add(param, alpha = wd) on the clipped gradient
        const float m0 = m[i];
        const float mi = __fmaf_rn(omb1, d - m0, m0);                       // exp_avg.lerp_(grad, 1 - beta1): m + (1 - b1)(g - m)
        const float vi = __fmaf_rn(omb2 * d, d, b2 * v[i]);                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bias2_sqrt + eps;
        p[i] = w - step_size * (mi / denom);                                // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
}

static int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace pemp

using namespace pemp;

static inline int nchunks_rows(int M) { return cdiv(M, RCHUNK); }

extern "C" size_t pemp_colsum_workspace_bytes(int M, int C) {
    return (size_t)nchunks_rows(M) * 2 * C * sizeof(double);
}

#define CHK_VEC(ptr, ld, C, what)                                                                         \
    PEMP_REQUIRE((ptr) && (C) % 4 == 0 && (ld) % 4 == 0 && (ld) >= (C) && (((uintptr_t)(ptr)) & 15) == 0, \
                 what ": pointer must be 16-byte aligned and C/ld multiples of 4")

extern "C" int pemp_bn_stats_f32(const float* z, int ldz, int M, int C, float eps, float momentum, float* mean,
                                 float* invstd, float* run_mean, float* run_var, void* ws, size_t ws_bytes,
                                 void* stream) {
    CHK_VEC(z, ldz, C, "bn_stats");
    PEMP_REQUIRE(M > 0 && mean && invstd && ws && ws_bytes >= pemp_colsum_workspace_bytes(M, C), "bn_stats: bad arguments");
    PEMP_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "bn_stats: running stats must both be given or both NULL");
    const int nck = nchunks_rows(M);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_kernel<0>, dim3(nck, cdiv(C, 64)), dim3(256), 0, st, z, ldz, (const float*)nullptr, 0,
                       (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, (double*)ws, M, C, 0);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, (const double*)ws, nck, M, C, eps,
                       momentum, mean, invstd, run_mean, run_var);
    return launch_status("bn_stats");
}

extern "C" int pemp_bn_stats_partials_f32(const float* stats, int nrows, int M, int C, float eps, float momentum, float* mean,
                                          float* invstd, float* run_mean, float* run_var, void* stream) {
    PEMP_REQUIRE(stats && mean && invstd && M > 0 && C > 0 && C % 32 == 0 && nrows > 0, "bn_stats_partials: bad arguments (C %% 32)");
    PEMP_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "bn_stats_partials: running stats must both be given or both NULL");
    hipLaunchKernelGGL(bn_stats_partials_kernel, dim3(C / 32), dim3(256), 0, (hipStream_t)stream, stats, nrows, M, C, eps,
                       momentum, mean, invstd, run_mean, run_var);
    return launch_status("bn_stats_partials");
}

extern "C" int pemp_bn_apply_f32(const float* z, int ldz, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, const float* residual, int ldr, float* y, int ldy, int M, int C,
                                 int relu, void* stream) {
    return pemp_bn_apply_mask_f32(z, ldz, mean, invstd, gamma, beta, residual, ldr, y, ldy, M, C, relu, nullptr, stream);
}

extern "C" int pemp_bn_apply_mask_f32(const float* z, int ldz, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, const float* residual, int ldr, float* y, int ldy, int M, int C,
                                      int relu, uint32_t* mask, void* stream) {
    CHK_VEC(z, ldz, C, "bn_apply");
    CHK_VEC(y, ldy, C, "bn_apply");
    PEMP_REQUIRE(M > 0 && mean && invstd && gamma && beta, "bn_apply: null pointer");
    if (residual) CHK_VEC(residual, ldr, C, "bn_apply");
    PEMP_REQUIRE(!mask || (C % 32 == 0 && C <= 1024 && 256 % (C / 4) == 0), "bn_apply: the sign mask needs C in {32, 64, 128, 256, 512, 1024}");
    const long long total = (long long)M * (C / 4);
    if (rows_form(C) || mask)
        hipLaunchKernelGGL(bn_apply_rows_kernel, dim3(rows_grid(M, C)), dim3(256), 0, (hipStream_t)stream, z, ldz, mean, invstd,
                           gamma, beta, residual, ldr, y, ldy, M, C / 4, relu, mask);
    else
        hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, ldz, mean,
                           invstd, gamma, beta, residual, ldr, y, ldy, (long long)M, C / 4, relu);
    return launch_status("bn_apply");
}

// BatchNorm apply (no residual, no ReLU) + the DropBlock2D behind it in one pass (ASPPV2's BN -> DropBlock -> conv branches)
extern "C" int pemp_bn_apply_dropblock_f32(const float* z, int ldz, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, float* y, int ldy, int M, int C, const float* rowmask,
                                           const int* kept_count, void* stream) {
    CHK_VEC(z, ldz, C, "bn_apply_dropblock");
    CHK_VEC(y, ldy, C, "bn_apply_dropblock");
    PEMP_REQUIRE(M > 0 && mean && invstd && gamma && beta && rowmask && kept_count, "bn_apply_dropblock: null pointer");
    PEMP_REQUIRE(rows_form(C), "bn_apply_dropblock: C must be one of the row-kernel widths (32 .. 1024, a power of two)");
    hipLaunchKernelGGL(bn_apply_rows_kernel, dim3(rows_grid(M, C)), dim3(256), 0, (hipStream_t)stream, z, ldz, mean, invstd,
                       gamma, beta, (const float*)nullptr, 0, y, ldy, M, C / 4, 0, (uint32_t*)nullptr, rowmask, kept_count);
    return launch_status("bn_apply_dropblock");
}

extern "C" int pemp_bn_bwd_f32(const float* dy, int lddy, const float* y, int ldy, const float* z, int ldz,
                               const float* mean, const float* invstd, const float* gamma, float* dz, int lddz,
                               float* gout, int ldg, float* dgamma, float* dbeta, int M, int C, int relu, void* ws,
                               size_t ws_bytes, void* stream) {
    return pemp_bn_bwd_mask_f32(dy, lddy, y, ldy, nullptr, z, ldz, mean, invstd, gamma, dz, lddz, gout, ldg, dgamma, dbeta, M, C, relu,
                                ws, ws_bytes, stream);
}

extern "C" int pemp_bn_bwd_mask_f32(const float* dy, int lddy, const float* y, int ldy, const uint32_t* mask, const float* z,
                                    int ldz, const float* mean, const float* invstd, const float* gamma, float* dz, int lddz,
                                    float* gout, int ldg, float* dgamma, float* dbeta, int M, int C, int relu, void* ws,
                                    size_t ws_bytes, void* stream) {
    CHK_VEC(dy, lddy, C, "bn_bwd");
    CHK_VEC(z, ldz, C, "bn_bwd");
    CHK_VEC(dz, lddz, C, "bn_bwd");
    PEMP_REQUIRE(!mask || (C % 32 == 0 && rows_form(C)), "bn_bwd: the sign mask needs C in {32, 64, 128, 256, 512, 1024}");
    if (relu && !mask) CHK_VEC(y, ldy, C, "bn_bwd");
    if (gout) CHK_VEC(gout, ldg, C, "bn_bwd");
    PEMP_REQUIRE(M > 0 && mean && invstd && gamma && dgamma && dbeta && ws, "bn_bwd: null pointer");
    PEMP_REQUIRE(ws_bytes >= pemp_colsum_workspace_bytes(M, C), "bn_bwd: workspace too small");
    const int nck = nchunks_rows(M);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_kernel<1>, dim3(nck, cdiv(C, 64)), dim3(256), 0, st, dy, lddy, y, ldy, z, ldz, mean, invstd,
                       (double*)ws, M, C, relu, mask);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, (const double*)ws, nck, C, dbeta, dgamma);
    const long long total = (long long)M * (C / 4);
    if (rows_form(C))
        hipLaunchKernelGGL(bn_bwd_apply_rows_kernel, dim3(rows_grid(M, C)), dim3(256), 0, st, dy, lddy, y, ldy, z, ldz, mean,
                           invstd, gamma, dbeta, dgamma, dz, lddz, gout, ldg, M, C / 4, relu, mask);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, dy, lddy, y, ldy, z, ldz, mean,
                           invstd, gamma, dbeta, dgamma, dz, lddz, gout, ldg, (long long)M, C / 4, relu);
    return launch_status("bn_bwd");
}

extern "C" int pemp_bn_fwd_partials_f32(const float* z, int ldz, const float* stats, int nrows, int M, int C, float eps,
                                        float momentum, const float* gamma, const float* beta, const float* residual, int ldr,
                                        float* y, int ldy, int relu, uint32_t* mask, float* mean, float* invstd, float* run_mean,
                                        float* run_var, void* stream) {
    CHK_VEC(z, ldz, C, "bn_fwd_partials");
    CHK_VEC(y, ldy, C, "bn_fwd_partials");
    if (residual) CHK_VEC(residual, ldr, C, "bn_fwd_partials");
    PEMP_REQUIRE(stats && nrows > 0 && M > 0 && gamma && beta && mean && invstd, "bn_fwd_partials: null pointer");
    PEMP_REQUIRE(C % 32 == 0, "bn_fwd_partials: C %% 32");
    PEMP_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "bn_fwd_partials: running stats must both be given or both NULL");
    // (one launch -- every block adding the partial rows of its 32 channels itself before it normalises its rows -- was
    // built and measured slower: 30.7 us against 24 us for the pair, the blocks wait out the partials' latency first)
    const int rc = pemp_bn_stats_partials_f32(stats, nrows, M, C, eps, momentum, mean, invstd, run_mean, run_var, stream);
    if (rc) return rc;
    return pemp_bn_apply_mask_f32(z, ldz, mean, invstd, gamma, beta, residual, ldr, y, ldy, M, C, relu, mask, stream);
}

extern "C" int pemp_bn_bwd_partials_f32(const float* g, int ldg, const float* z, int ldz, const float* mean,
                                        const float* invstd, const float* gamma, const float* stats, int nrows, float* dz,
                                        int lddz, float* dgamma, float* dbeta, int M, int C, void* stream) {
    CHK_VEC(g, ldg, C, "bn_bwd_partials");
    CHK_VEC(z, ldz, C, "bn_bwd_partials");
    CHK_VEC(dz, lddz, C, "bn_bwd_partials");
    PEMP_REQUIRE(M > 0 && mean && invstd && gamma && dgamma && dbeta && stats && nrows > 0, "bn_bwd_partials: null pointer");
    PEMP_REQUIRE(C % 32 == 0, "bn_bwd_partials: C %% 32");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(C / 32), dim3(256), 0, st, stats, nrows, C, dbeta, dgamma);
    const long long total = (long long)M * (C / 4);
    if (rows_form(C))
        hipLaunchKernelGGL(bn_bwd_apply_rows_kernel, dim3(rows_grid(M, C)), dim3(256), 0, st, g, ldg, (const float*)nullptr, 0, z,
                           ldz, mean, invstd, gamma, dbeta, dgamma, dz, lddz, (float*)nullptr, 0, M, C / 4, 0);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, g, ldg, (const float*)nullptr, 0, z,
                           ldz, mean, invstd, gamma, dbeta, dgamma, dz, lddz, (float*)nullptr, 0, (long long)M, C / 4, 0);
    return launch_status("bn_bwd_partials");
}

extern "C" int pemp_relu_bias_bwd_f32(const float* dy, int lddy, const float* y, int ldy, const float* add, int lda,
                                      float* g, int ldg, float* dbias, int M, int C, int relu, void* ws,
                                      size_t ws_bytes, void* stream) {
    CHK_VEC(dy, lddy, C, "relu_bias_bwd");
    CHK_VEC(g, ldg, C, "relu_bias_bwd");
    if (relu) CHK_VEC(y, ldy, C, "relu_bias_bwd");
    if (add) CHK_VEC(add, lda, C, "relu_bias_bwd");
    PEMP_REQUIRE(M > 0, "relu_bias_bwd: M <= 0");
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)M * (C / 4);
    if (rows_form(C))
        hipLaunchKernelGGL(relu_bwd_rows_kernel, dim3(rows_grid(M, C)), dim3(256), 0, st, dy, lddy, y, ldy, add, lda, g, ldg,
                           M, C / 4, relu);
    else
        hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, dy, lddy, y, ldy, add, lda, g, ldg,
                           (long long)M, C / 4, relu);
    if (dbias) {
        PEMP_REQUIRE(ws && ws_bytes >= pemp_colsum_workspace_bytes(M, C), "relu_bias_bwd: workspace too small");
        const int nck = nchunks_rows(M);
        hipLaunchKernelGGL(colsum_kernel<2>, dim3(nck, cdiv(C, 64)), dim3(256), 0, st, (const float*)g, ldg,
                           (const float*)nullptr, 0, (const float*)nullptr, 0, (const float*)nullptr,
                           (const float*)nullptr, (double*)ws, M, C, 0);
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, (const double*)ws, nck, C, dbias,
                           (float*)nullptr);
    }
    return launch_status("relu_bias_bwd");
}

extern "C" int pemp_maxpool2d_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int N, int H, int W, int C,
                                           int Ho, int Wo, int k, int s, int p, void* stream) {
    PEMP_REQUIRE(x && dy && dx && N > 0 && C % 4 == 0, "maxpool_bwd: bad arguments");
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H,
                       W, C / 4, Ho, Wo, k, s, p);
    return launch_status("maxpool_bwd");
}

extern "C" int pemp_scatter_strided_nhwc_f32(const float* src, float* dst, int N, int H, int W, int Hs, int Ws, int C,
                                             int s, void* stream) {
    PEMP_REQUIRE(src && dst && N > 0 && C % 4 == 0 && s >= 1, "scatter_strided: bad arguments");
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(scatter_strided_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, N,
                       H, W, Hs, Ws, C / 4, s);
    return launch_status("scatter_strided");
}

// table: device int32 [L][6] = {offset (floats), Cout, taps, Cin, first tile, tiles along Cin} per conv weight, layers in
// ascending tile order; total_tiles = sum over layers of taps * ceil(Cout / 32) * ceil(Cin / 32).
extern "C" int pemp_dgrad_mirror_f32(const float* params, float* mirror, const int32_t* table, int L, int total_tiles,
                                     void* stream) {
    PEMP_REQUIRE(params && mirror && table && L > 0 && total_tiles > 0, "dgrad_mirror: bad arguments");
    hipLaunchKernelGGL(dgrad_mirror_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, params, mirror, (const int*)table, L);
    return launch_status("dgrad_mirror");
}

extern "C" int pemp_gap_bwd_add_nhwc_f32(const float* v, float* dst, int ld, int N, int HW, int C, void* stream) {
    PEMP_REQUIRE(v && dst && N > 0 && HW > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "gap_bwd: bad arguments");
    const long long total = (long long)N * HW * (C / 4);
    hipLaunchKernelGGL(gap_bwd_add_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, v, dst, ld, HW,
                       C / 4, total);
    return launch_status("gap_bwd");
}

extern "C" size_t pemp_sgd_workspace_bytes(void) { return 1024 * sizeof(double) + 16; }

// grad_norm_out[0] = ||g||_2 ; then p/buf updated.  max_norm <= 0 disables clipping.  grad_scale
// multiplies every gradient first (1/world after a SUM all-reduce).
extern "C" int pemp_sgd_clip_step_f32(float* params, const float* grads, float* momentum_buf, long long n,
                                      float max_norm, float lr, float momentum, float weight_decay, int first_step,
                                      float grad_scale, int nesterov, float* grad_norm_out, void* ws, size_t ws_bytes,
                                      void* stream) {
    PEMP_REQUIRE(params && grads && momentum_buf && grad_norm_out && ws && n > 0, "sgd_clip_step: bad arguments");
    PEMP_REQUIRE(ws_bytes >= pemp_sgd_workspace_bytes(), "sgd_clip_step: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = 1024;
    hipLaunchKernelGGL(sqsum_partial_kernel, dim3(nb), dim3(256), 0, st, grads, n, (double*)ws);
    hipLaunchKernelGGL(sqsum_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nb, grad_norm_out);
    hipLaunchKernelGGL(sgd_clip_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, params, grads, momentum_buf, n,
                       (const float*)grad_norm_out, max_norm, lr, momentum, weight_decay, first_step, grad_scale, nesterov);
    return launch_status("sgd_clip_step");
}

// clip_grad_norm_(max_norm) + torch.optim.Adam(lr, betas, eps, weight_decay).step() on the flat buffers (reference
// core/solver.py:92-96: tr.opt = adam).  step = the 1-based number of this update (bias corrections 1 - beta^step).
extern "C" int pemp_adam_clip_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                                       float max_norm, double lr, double beta1, double beta2, double eps, double weight_decay,
                                       long long step, float grad_scale, float* grad_norm_out, void* ws, size_t ws_bytes,
                                       void* stream) {
    PEMP_REQUIRE(params && grads && exp_avg && exp_avg_sq && grad_norm_out && ws && n > 0 && step >= 1, "adam_clip_step: bad arguments");
    PEMP_REQUIRE(ws_bytes >= pemp_sgd_workspace_bytes(), "adam_clip_step: workspace too small");
    PEMP_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "adam_clip_step: betas in [0, 1), eps >= 0");
    hipStream_t st = (hipStream_t)stream;
    const int nb = 1024;
    hipLaunchKernelGGL(sqsum_partial_kernel, dim3(nb), dim3(256), 0, st, grads, n, (double*)ws);
    hipLaunchKernelGGL(sqsum_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nb, grad_norm_out);
    // torch/optim/adam.py: bias corrections, step_size and sqrt(bias2) are Python-float (double) arithmetic on the host
    const double bias1 = 1.0 - pow(beta1, (double)step), bias2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_clip_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n,
                       (const float*)grad_norm_out, max_norm, (float)(lr / bias1), (float)sqrt(bias2), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, grad_scale);
    return launch_status("adam_clip_step");
}
