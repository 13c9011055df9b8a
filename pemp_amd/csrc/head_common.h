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
//   A[BS][J][n] | part[BS][nchunks][J][c] | asum[BS][nchunks][J] | msum[BS][2]
struct PoolWs {
    float* A;
    float* part;
    float* asum;
    float* msum;
};
static inline size_t pool_ws_floats(int BS, int n, int c, int J) {
    const size_t nck = nchunks_of(n);
    return (size_t)BS * J * n + (size_t)BS * nck * J * c + (size_t)BS * nck * J + (size_t)BS * 2 + 16;
}
static inline PoolWs pool_ws_layout(void* ws, int BS, int n, int c, int J) {
    const size_t nck = nchunks_of(n);
    PoolWs l;
    l.A = (float*)ws;
    l.part = l.A + (size_t)BS * J * n;
    l.asum = l.part + (size_t)BS * nck * J * c;
    l.msum = l.asum + (size_t)BS * nck * J;
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
