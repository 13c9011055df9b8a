// Shared between head.hip (forward) and head_bwd.hip (backward): limits, the pooling workspace layout
// and the index helpers that both directions must evaluate identically.
#pragma once
#include "common.h"

namespace pemp {

// MAXJ (the rows per pixel, 2 * protos, a kernel keeps in registers) is set per instantiation of the head bodies: head.hip
constexpr int MAXCL = 8;    // channels per lane: c <= 64*MAXCL = 512
constexpr int PCHUNK = 64;  // pixels per pooling block

static inline int nchunks_of(int n) { return cdiv(n, PCHUNK); }

// workspace of pemp_mpm_protos_f32 / pemp_masked_avg_pool_f32 (all fp32):
//   A[BS][J][n] | part[BS][nchunks][J][c] | asum[BS][nchunks][J] | msum[BS][2] | dc[c][J] | cm[2][c] | bias[J]
// (the last three: the centred MPM centres, see centre_kernel below)
struct PoolWs {
    float* A;
    float* part;
    float* asum;
    float* msum;
    float* dc;
    float* cm;
    float* bias;
};
static inline size_t pool_ws_floats(int BS, int n, int c, int J) {
    const size_t nck = nchunks_of(n);
    return (size_t)BS * J * n + (size_t)BS * nck * J * c + (size_t)BS * nck * J + (size_t)BS * 2 + 16 + (size_t)J * c +
           2 * (size_t)c + J + 16;
}
static inline PoolWs pool_ws_layout(void* ws, int BS, int n, int c, int J) {
    const size_t nck = nchunks_of(n);
    PoolWs l;
    l.A = (float*)ws;
    l.part = l.A + (size_t)BS * J * n;
    l.asum = l.part + (size_t)BS * nck * J * c;
    l.msum = l.asum + (size_t)BS * nck * J;
    l.dc = l.msum + (size_t)BS * 2 + 16;
    l.cm = l.dc + (size_t)J * c;
    l.bias = l.cm + 2 * (size_t)c;
    return l;
}

// Sum over the pooling chunks of one (image, row j) by a 256-thread block: wave v adds chunks v, v+4, ... in
// order, the four partial sums combine as (s0 + s1) + (s2 + s3) -- one order for the forward (pool_final) and the
// backward (pool_shot).  `p` already carries the lane's channel offset; every thread of the block must call.
__device__ __forceinline__ float chunk_sum(const float* __restrict__ p, size_t stride, int nchunks, float (*red)[64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
#pragma unroll 4
    for (int k = wave; k < nchunks; k += 4) s += p[k * stride];
    red[wave][lane] = s;
    __syncthreads();
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    __syncthreads();
    return s;
}

// Shift-invariant form of the MPM soft assignment (networks/pemp_stage1.py:205-207).  The reference takes
// softmax_j(-|x - c_j|^2) inside the foreground and the background group of centres.  In float32 the squared distances are
// sums of c terms that reach the hundreds, their rounding (~1e-4 absolute) goes straight into the softmax, and the
// gradients of a training step inherit a 1e-4 relative error -- in the reference's own float32 run as well (measured against
// float64: tests/test_grad_frozen_gpu.py).  A softmax only sees differences inside its group, and
//     -|x - c_j|^2 + |x - m_g|^2  =  2 x . (c_j - m_g) - (|c_j|^2 - |m_g|^2)        for ANY vector m_g,
// so the kernels work with the centres relative to their group mean m_g: the products x . (c_j - m_g) are an order of
// magnitude smaller than the distances, and the bias (c_j - m_g) . (c_j + m_g) is accumulated in double.  Same function,
// 20x closer to its float64 value (1e-5 instead of 1e-4 in the step's gradients).  centre_kernel (one 256-thread block
// ahead of the assignment) leaves, in the tail of the pooling workspace,
//   cm[g][ch]   = (c_{g,0}[ch] + c_{g,1}[ch] + ...) / (float)p      ascending j, float32
//   dc[ch][j]   = c_j[ch] - cm[g(j)][ch]                             float32, laid out like ctr ([c][2p])
//   bias[j]     = float( sum_ch (double)dc * ((double)dc + 2 (double)cm) )      fixed order
// and the forward (both variants) and the backward read them from there, so they see the same assignment.
template <int MJ>
__global__ __launch_bounds__(256) void centre_kernel(const float* __restrict__ ctr, float* __restrict__ dc,
                                                     float* __restrict__ cm, float* __restrict__ bias, int c, int p) {
    __shared__ double red[MJ][4];
    const int J = 2 * p, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) s[j] = 0.0;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float cv[MJ], s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            cv[j] = j < J ? ctr[ch * J + j] : 0.f;
            if (j < p) s0 += cv[j];
            else if (j < J) s1 += cv[j];
        }
        const float m0 = s0 / (float)p, m1 = s1 / (float)p;
        cm[ch] = m0;
        cm[c + ch] = m1;
#pragma unroll
        for (int j = 0; j < MJ; ++j)
            if (j < J) {
                const float mg = j < p ? m0 : m1, d = cv[j] - mg;
                dc[ch * J + j] = d;
                s[j] += (double)d * ((double)d + 2.0 * (double)mg);
            }
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        double v = s[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[j][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < J) bias[threadIdx.x] = (float)((red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

// F.interpolate(mode="nearest") source index (legacy rule: floor(dst * in/out), scale in fp32)
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    float scale = (float)in / (float)out;
    int s = (int)floorf((float)dst * scale);
    return min(s, in - 1);
}

// bilinear, align_corners=True (ATen area_pixel_compute_scale: (in-1)/(out-1) in fp32)
struct Bilin {
    int i0, i1;
    float l;
};
__device__ __forceinline__ Bilin bilin(int dst, int in, int out) {
    float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float f = scale * (float)dst;
    Bilin b;
    b.i0 = min((int)f, in - 1);
    b.i1 = b.i0 + (b.i0 < in - 1 ? 1 : 0);
    b.l = f - (float)b.i0;
    return b;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ p, int w, Bilin by, Bilin bx) {
    float v00 = p[by.i0 * w + bx.i0], v01 = p[by.i0 * w + bx.i1];
    float v10 = p[by.i1 * w + bx.i0], v11 = p[by.i1 * w + bx.i1];
    // explicit rounding points so that every kernel using this helper produces the same bits
    float h0 = 1.f - by.l, w0 = 1.f - bx.l;
    float top = __fmaf_rn(bx.l, v01, __fmul_rn(w0, v00));
    float bot = __fmaf_rn(bx.l, v11, __fmul_rn(w0, v10));
    return __fmaf_rn(by.l, bot, __fmul_rn(h0, top));
}

}  // namespace pemp
