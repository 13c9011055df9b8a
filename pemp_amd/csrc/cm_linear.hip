// The small linear algebra of ResNetCM.comm (reference networks/backbones.py:213-221) and its backward:
// episode mean of the per-image (mean, max) statistics, Linear(2C -> 2), broadcast back per image, and the
// per-image bias the two constant channels contribute to the first 1x1 convs of a stage.  Sizes are tiny
// (N <= a few hundred images, 2C <= 2048): latency-bound, one short launch each, every sum in a fixed order.
#include "common.h"

namespace pemp {

constexpr int CMG = 64;      // max episodes per call

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// one block per episode g: agg[g][k] = mean_{n in g} stat[n][k];  feat[g][e] = b[e] + sum_k agg[g][k] * W[e][k]
__global__ __launch_bounds__(256) void cm_linear_kernel(const float* __restrict__ stat, const int* __restrict__ group,
                                                        const float* __restrict__ W, const float* __restrict__ b,
                                                        float* __restrict__ agg, float* __restrict__ feat, int N, int C2) {
    __shared__ float red[256];
    const int g = blockIdx.x;
    int cnt = 0;
    for (int n = 0; n < N; ++n) cnt += group[n] == g;
    float p0 = 0.f, p1 = 0.f;
    for (int k = threadIdx.x; k < C2; k += 256) {
        float s = 0.f;
        for (int n = 0; n < N; ++n)
            if (group[n] == g) s += stat[(size_t)n * C2 + k];
        const float a = cnt > 0 ? s / (float)cnt : 0.f;
        agg[(size_t)g * C2 + k] = a;
        p0 += a * W[k];
        p1 += a * W[C2 + k];
    }
    const float s0 = block_sum_256(p0, red), s1 = block_sum_256(p1, red);
    if (threadIdx.x == 0) {
        feat[g * 2 + 0] = b[0] + s0;
        feat[g * 2 + 1] = b[1] + s1;
    }
}

__global__ void cm_bias_kernel(const float* __restrict__ feat, const int* __restrict__ group, const float* __restrict__ wext,
                               int ldw, const float* __restrict__ alpha, const float* __restrict__ base,
                               float* __restrict__ out, int N, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Cout) return;
    const int n = i / Cout, co = i - n * Cout;
    const float* f = feat + group[n] * 2;
    const float d = f[0] * wext[(size_t)co * ldw] + f[1] * wext[(size_t)co * ldw + 1];
    out[i] = (base ? base[co] : 0.f) + (alpha ? alpha[co] * d : d);
}

// dwext[co][e] = sum_n colsum[n][co] * feat[g(n)][e]          (thread per co, images in index order)
__global__ void cm_bias_bwd_w_kernel(const float* __restrict__ colsum, const float* __restrict__ feat,
                                     const int* __restrict__ group, float* __restrict__ dwext, int lddw, int N, int Cout) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= Cout) return;
    float d0 = 0.f, d1 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float c = colsum[(size_t)n * Cout + co];
        d0 += c * feat[group[n] * 2];
        d1 += c * feat[group[n] * 2 + 1];
    }
    dwext[(size_t)co * lddw] = d0;
    dwext[(size_t)co * lddw + 1] = d1;
}

// dfeat_img[n][e] (+)= sum_co colsum[n][co] * wext[co][e]       (block per image)
__global__ __launch_bounds__(256) void cm_bias_bwd_f_kernel(const float* __restrict__ colsum, const float* __restrict__ wext, int ldw,
                                                            float* __restrict__ dfi, int accumulate, int Cout) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    float p0 = 0.f, p1 = 0.f;
    for (int co = threadIdx.x; co < Cout; co += 256) {
        const float c = colsum[(size_t)n * Cout + co];
        p0 += c * wext[(size_t)co * ldw];
        p1 += c * wext[(size_t)co * ldw + 1];
    }
    const float s0 = block_sum_256(p0, red), s1 = block_sum_256(p1, red);
    if (threadIdx.x == 0) {
        dfi[n * 2 + 0] = (accumulate ? dfi[n * 2 + 0] : 0.f) + s0;
        dfi[n * 2 + 1] = (accumulate ? dfi[n * 2 + 1] : 0.f) + s1;
    }
}

// Linear + episode mean backward.  Every block first rebuilds dfeat[g][e] = sum_{n in g} dfi[n][e] and the episode
// sizes in LDS; then thread k: dW[e][k] = sum_g dfeat[g][e] agg[g][k];  dstat[n][k] = (sum_e dfeat[g][e] W[e][k]) / cnt_g.
__global__ __launch_bounds__(256) void cm_linear_bwd_kernel(const float* __restrict__ dfi, const int* __restrict__ group,
                                                            const float* __restrict__ agg, const float* __restrict__ W,
                                                            float* __restrict__ dW, float* __restrict__ db,
                                                            float* __restrict__ dstat, int N, int G, int C2) {
    __shared__ float df[CMG][2];
    __shared__ int cnt[CMG];
    if ((int)threadIdx.x < G) {
        const int g = threadIdx.x;
        float a = 0.f, b = 0.f;
        int c = 0;
        for (int n = 0; n < N; ++n)
            if (group[n] == g) {
                a += dfi[n * 2];
                b += dfi[n * 2 + 1];
                ++c;
            }
        df[g][0] = a;
        df[g][1] = b;
        cnt[g] = c;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += df[g][threadIdx.x];
        db[threadIdx.x] = s;
    }
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= C2) return;
    float w0 = 0.f, w1 = 0.f;
    for (int g = 0; g < G; ++g) {
        const float a = agg[(size_t)g * C2 + k];
        w0 += df[g][0] * a;
        w1 += df[g][1] * a;
    }
    dW[k] = w0;
    dW[C2 + k] = w1;
    const float W0 = W[k], W1 = W[C2 + k];
    for (int n = 0; n < N; ++n) {
        const int g = group[n];
        dstat[(size_t)n * C2 + k] = (df[g][0] * W0 + df[g][1] * W1) / (float)cnt[g];
    }
}

}  // namespace pemp

using namespace pemp;

extern "C" int pemp_cm_linear_f32(const float* stat, const int32_t* group, const float* lin_w, const float* lin_b, float* agg,
                                  float* feat, int N, int G, int C2, void* stream) {
    PEMP_REQUIRE(stat && group && lin_w && lin_b && agg && feat, "cm_linear: null pointer");
    PEMP_REQUIRE(N > 0 && G > 0 && G <= CMG && C2 > 0, "cm_linear: bad dims (episodes per call <= %d)", CMG);
    hipLaunchKernelGGL(cm_linear_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, stat, group, lin_w, lin_b, agg, feat, N, C2);
    return launch_status("cm_linear");
}

extern "C" int pemp_cm_bias_f32(const float* feat, const int32_t* group, const float* wext, int ldw, const float* alpha,
                                const float* base, float* out, int N, int Cout, void* stream) {
    PEMP_REQUIRE(feat && group && wext && out && N > 0 && Cout > 0 && ldw >= 2, "cm_bias: bad arguments");
    hipLaunchKernelGGL(cm_bias_kernel, dim3(cdiv(N * Cout, 256)), dim3(256), 0, (hipStream_t)stream, feat, group, wext, ldw, alpha,
                       base, out, N, Cout);
    return launch_status("cm_bias");
}

extern "C" int pemp_cm_bias_bwd_f32(const float* colsum, const float* feat, const int32_t* group, const float* wext, int ldw,
                                    float* dwext, int lddw, float* dfeat_img, int accumulate, int N, int Cout, void* stream) {
    PEMP_REQUIRE(colsum && feat && group && wext && dwext && dfeat_img && N > 0 && Cout > 0 && ldw >= 2 && lddw >= 2,
                 "cm_bias_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cm_bias_bwd_w_kernel, dim3(cdiv(Cout, 256)), dim3(256), 0, st, colsum, feat, group, dwext, lddw, N, Cout);
    hipLaunchKernelGGL(cm_bias_bwd_f_kernel, dim3(N), dim3(256), 0, st, colsum, wext, ldw, dfeat_img, accumulate, Cout);
    return launch_status("cm_bias_bwd");
}

extern "C" int pemp_cm_linear_bwd_f32(const float* dfeat_img, const int32_t* group, const float* agg, const float* lin_w,
                                      float* dlin_w, float* dlin_b, float* dstat, int N, int G, int C2, void* stream) {
    PEMP_REQUIRE(dfeat_img && group && agg && lin_w && dlin_w && dlin_b && dstat, "cm_linear_bwd: null pointer");
    PEMP_REQUIRE(N > 0 && G > 0 && G <= CMG && C2 > 0, "cm_linear_bwd: bad dims");
    hipLaunchKernelGGL(cm_linear_bwd_kernel, dim3(cdiv(C2, 256)), dim3(256), 0, (hipStream_t)stream, dfeat_img, group, agg, lin_w,
                       dlin_w, dlin_b, dstat, N, G, C2);
    return launch_status("cm_linear_bwd");
}
