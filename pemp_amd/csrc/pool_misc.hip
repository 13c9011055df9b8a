// Streaming (HBM/L2-bound) helper kernels around the conv engine: input packing, max pooling,
// global average pooling, multi-branch channel affine, ResNetCM statistics.  All NHWC, one
// float4 (16 B) per lane wherever the channel count allows.
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace pemp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------
__global__ void pack_input_kernel(const float* __restrict__ img, const float* __restrict__ prior,
                                  float4* __restrict__ out, int HW, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long n = i / HW;
        int p = (int)(i - n * HW);
        const float* b = img + n * 3 * HW + p;
        float4 v;
        v.x = b[0];
        v.y = b[HW];
        v.z = b[2 * (long long)HW];
        v.w = prior ? prior[n * HW + p] : 0.f;
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// max pool, one thread per (pixel, 4 channels); window clipped to the input (padding never wins
// because it is -inf in nn.MaxPool2d).
__global__ void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W,
                               int C4, int ldx, int Ho, int Wo, int ldy, int k, int s, int p) {
    long long total = (long long)N * Ho * Wo * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4);
        long long t = i / C4;
        int wo = (int)(t % Wo);
        t /= Wo;
        int ho = (int)(t % Ho);
        int n = (int)(t / Ho);
        int h0 = ho * s - p, w0 = wo * s - p;
        int h1 = min(h0 + k, H), w1 = min(w0 + k, W);
        h0 = max(h0, 0);
        w0 = max(w0, 0);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                float4 v = *(const float4*)(x + ((long long)(n * H + h) * W + w) * ldx + c4 * 4);
                m.x = fmaxf(m.x, v.x);
                m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z);
                m.w = fmaxf(m.w, v.w);
            }
        *(float4*)(y + ((long long)(n * Ho + ho) * Wo + wo) * ldy + c4 * 4) = m;
    }
}

// Training variant: also records WHICH window element won (first maximum in scan order, as ATen), as its
// offset (dh * k + dw) inside the unclipped k x k window -- one byte per output element -- so that the
// backward pass never has to re-scan the input.
__global__ void maxpool_idx_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                   int N, int H, int W, int C4, int Ho, int Wo, int k, int s, int p) {
    long long total = (long long)N * Ho * Wo * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4);
        long long t = i / C4;
        int wo = (int)(t % Wo);
        t /= Wo;
        int ho = (int)(t % Ho);
        int n = (int)(t / Ho);
        const int hb = ho * s - p, wb = wo * s - p;
        const int h0 = max(hb, 0), w0 = max(wb, 0), h1 = min(hb + k, H), w1 = min(wb + k, W);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int pos[4] = {-1, -1, -1, -1};
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                const float4 v4 = *(const float4*)(x + (((long long)n * H + h) * W + w) * C4 * 4 + c4 * 4);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > m[e] || pos[e] < 0) {
                        m[e] = v[e];
                        pos[e] = (h - hb) * k + (w - wb);
                    }
            }
        *(float4*)(y + i * 4) = make_float4(m[0], m[1], m[2], m[3]);
        *(uchar4*)(idx + i * 4) = make_uchar4((uint8_t)pos[0], (uint8_t)pos[1], (uint8_t)pos[2], (uint8_t)pos[3]);
    }
}
// dx[pixel] = sum of dy over the (at most ceil(k/s)^2) windows whose recorded winner is this pixel
__global__ void maxpool_idx_bwd_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dy,
                                       float* __restrict__ dx, int N, int H, int W, int C4, int Ho, int Wo, int k, int s,
                                       int p) {
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int ho_lo = max(0, (h + p - k + s) / s), ho_hi = min(Ho - 1, (h + p) / s);
        const int wo_lo = max(0, (w + p - k + s) / s), wo_hi = min(Wo - 1, (w + p) / s);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const long long o = ((((long long)n * Ho + ho) * Wo + wo) * C4 + c4) * 4;
                const uchar4 q = *(const uchar4*)(idx + o);
                const int me = (h - (ho * s - p)) * k + (w - (wo * s - p));
                if (q.x == me || q.y == me || q.z == me || q.w == me) {
                    const float4 g = *(const float4*)(dy + o);
                    acc[0] += q.x == me ? g.x : 0.f;
                    acc[1] += q.y == me ? g.y : 0.f;
                    acc[2] += q.z == me ? g.z : 0.f;
                    acc[3] += q.w == me ? g.w : 0.f;
                }
            }
        *(float4*)(dx + i * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// global average pool: block = (image, 64 channels); 1024 threads = 64 pixel lanes x 16 float4 channel
// lanes, so every load is 16 B and a wave reads 4 pixels x 256 contiguous bytes.  Fixed summation
// order (lane-strided partial sums, then a tree over the 64 lanes) -> deterministic.
__global__ __launch_bounds__(1024) void gap_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   int HW, int C, int ldx) {
    __shared__ float4 red[64][16];
    const int n = blockIdx.y;
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        const float* b = x + (long long)n * HW * ldx + c;
        for (int i = pl; i < HW; i += 64) {
            float4 v = *(const float4*)(b + (long long)i * ldx);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[pl][cl] = s;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        if (pl < o) {
            float4 a = red[pl][cl], b2 = red[pl + o][cl];
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            red[pl][cl] = a;
        }
        __syncthreads();
    }
    if (pl == 0 && c < C) {
        float4 t = red[0][cl];
        const float inv = (float)HW;
        *(float4*)(y + (long long)n * C + c) = make_float4(t.x / inv, t.y / inv, t.z / inv, t.w / inv);
    }
}

// ---------------------------------------------------------------------------------------------
struct AffineMulti {
    const float* scale[4];
    const float* shift[4];
    float* y[4];
    int ldy[4];
};

__global__ void affine_multi_kernel(const float* __restrict__ x, int ldx, long long M, int C4, int nb,
                                    AffineMulti a) {
    long long total = M * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4);
        long long m = i / C4;
        float4 v = *(const float4*)(x + m * ldx + c4 * 4);
        for (int b = 0; b < nb; ++b) {
            float4 sc = *(const float4*)(a.scale[b] + c4 * 4);
            float4 sh = *(const float4*)(a.shift[b] + c4 * 4);
            float4 o;
            o.x = v.x * sc.x + sh.x;
            o.y = v.y * sc.y + sh.y;
            o.z = v.z * sc.z + sh.z;
            o.w = v.w * sc.w + sh.w;
            *(float4*)(a.y[b] + m * a.ldy[b] + c4 * 4) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ResNetCM.comm statistics.  Kernel 1: mask' = max_pool2d(mask, 3, stride, 1).
__global__ void mask_pool_kernel(const float* __restrict__ mi, float* __restrict__ mo, int N, int Hm, int Wm,
                                 int Hx, int Wx, int s) {
    long long total = (long long)N * Hx * Wx;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int wo = (int)(i % Wx);
        long long t = i / Wx;
        int ho = (int)(t % Hx);
        int n = (int)(t / Hx);
        int h0 = max(ho * s - 1, 0), h1 = min(ho * s + 2, Hm);
        int w0 = max(wo * s - 1, 0), w1 = min(wo * s + 2, Wm);
        float m = -INFINITY;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) m = fmaxf(m, mi[((long long)n * Hm + h) * Wm + w]);
        mo[i] = m;
    }
}
// Kernel 2: per (image, 64-channel group): mean over all pixels and max over pixels of x*mask'.
__global__ __launch_bounds__(256) void cm_stat_kernel(const float* __restrict__ x, int ldx,
                                                      const float* __restrict__ mask, float* __restrict__ stat,
                                                      int HW, int C) {
    __shared__ float rs[4][64], rm[4][64];
    const int n = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;
    float s = 0.f, mx = -INFINITY;
    if (c < C) {
        const float* b = x + (long long)n * HW * ldx + c;
        const float* mk = mask + (long long)n * HW;
        for (int i = pl; i < HW; i += 4) {
            float v = b[(long long)i * ldx] * mk[i];
            s += v;
            mx = fmaxf(mx, v);
        }
    }
    rs[pl][threadIdx.x & 63] = s;
    rm[pl][threadIdx.x & 63] = mx;
    __syncthreads();
    if (pl == 0 && c < C) {
        int t = threadIdx.x;
        stat[((long long)n * 2 + 0) * C + c] = ((rs[0][t] + rs[1][t]) + (rs[2][t] + rs[3][t])) / (float)HW;
        stat[((long long)n * 2 + 1) * C + c] = fmaxf(fmaxf(rm[0][t], rm[1][t]), fmaxf(rm[2][t], rm[3][t]));
    }
}

// Adjoint of cm_stat_kernel: dx[n,i,c] += mask[n,i] * (dmean[n,c]/HW + [i == first argmax_i x*mask] * dmax[n,c]).
// Same block shape as the forward; the arg-max is recomputed (first index wins, as torch.max on the CPU does).
__global__ __launch_bounds__(256) void cm_bwd_kernel(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ mask, const float* __restrict__ dstat,
                                                     float* __restrict__ dx, int ldd, int HW, int C) {
    __shared__ float rm[4][64];
    __shared__ int ri[4][64];
    const int n = blockIdx.y;
    const int cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    const int pl = threadIdx.x >> 6;
    const float* mk = mask + (long long)n * HW;
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    if (c < C) {
        const float* b = x + (long long)n * HW * ldx + c;
        for (int i = pl; i < HW; i += 4) {
            float v = b[(long long)i * ldx] * mk[i];
            if (v > mx) {
                mx = v;
                arg = i;
            }
        }
    }
    rm[pl][cl] = mx;
    ri[pl][cl] = arg;
    __syncthreads();
    if (c >= C) return;
    for (int q = 0; q < 4; ++q) {
        float v = rm[q][cl];
        int a = ri[q][cl];
        if (q == 0 || v > mx || (v == mx && a < arg)) {
            mx = v;
            arg = a;
        }
    }
    const float gm = dstat[((long long)n * 2 + 0) * C + c] / (float)HW;
    const float gx = dstat[((long long)n * 2 + 1) * C + c];
    float* d = dx + (long long)n * HW * ldd + c;
    for (int i = pl; i < HW; i += 4) {
        float g = gm + (i == arg ? gx : 0.f);
        d[(long long)i * ldd] += mk[i] * g;
    }
}

// The same two kernels for C % 4 == 0 (every CM stage: 64 / 256 / 512 channels), shaped like gap_kernel: block =
// (image, 64 channels), 1024 threads = 64 pixel lanes x 16 float4 channel lanes, so a wave reads 4 pixels x 256
// contiguous bytes per load instead of 64 scalars (the scalar kernels above ran at 0.4 TB/s: 0.76 ms per call at 48
// images).  Fixed order: lane-strided partials, then a tree over the 64 pixel lanes -> deterministic.
// ARG: also record the index of the FIRST maximal pixel per (image, channel) (training: the backward pass then is a
// plain element-wise kernel, cm_bwd_arg_kernel).
template <bool ARG>
__global__ __launch_bounds__(1024) void cm_stat4_kernel(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ mask, float* __restrict__ stat,
                                                        int* __restrict__ argmax, int HW, int C) {
    __shared__ float4 rs[64][16], rm[64][16];
    __shared__ int4 ri[ARG ? 64 : 1][16];
    const int n = blockIdx.y;
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (c < C) {
        const float* b = x + (long long)n * HW * ldx + c;
        const float* mk = mask + (long long)n * HW;
        for (int i = pl; i < HW; i += 64) {
            const float4 v4 = *(const float4*)(b + (long long)i * ldx);
            const float k = mk[i];
            const float v[4] = {v4.x * k, v4.y * k, v4.z * k, v4.w * k};
            s.x += v[0]; s.y += v[1]; s.z += v[2]; s.w += v[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ARG) {
                    if (v[e] > mx[e]) {
                        mx[e] = v[e];
                        arg[e] = i;
                    }
                } else {
                    mx[e] = fmaxf(mx[e], v[e]);
                }
            }
        }
    }
    rs[pl][cl] = s;
    rm[pl][cl] = make_float4(mx[0], mx[1], mx[2], mx[3]);
    if (ARG) ri[pl][cl] = make_int4(arg[0], arg[1], arg[2], arg[3]);
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        if (pl < o) {
            float4 a = rs[pl][cl], b2 = rs[pl + o][cl];
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            rs[pl][cl] = a;
            float4 u = rm[pl][cl], w2 = rm[pl + o][cl];
            if (ARG) {                         // first (smallest-index) maximum wins, as torch.max on the CPU
                int4 iu = ri[pl][cl], iw = ri[pl + o][cl];
                if (w2.x > u.x || (w2.x == u.x && iw.x < iu.x)) { u.x = w2.x; iu.x = iw.x; }
                if (w2.y > u.y || (w2.y == u.y && iw.y < iu.y)) { u.y = w2.y; iu.y = iw.y; }
                if (w2.z > u.z || (w2.z == u.z && iw.z < iu.z)) { u.z = w2.z; iu.z = iw.z; }
                if (w2.w > u.w || (w2.w == u.w && iw.w < iu.w)) { u.w = w2.w; iu.w = iw.w; }
                ri[pl][cl] = iu;
            } else {
                u.x = fmaxf(u.x, w2.x); u.y = fmaxf(u.y, w2.y); u.z = fmaxf(u.z, w2.z); u.w = fmaxf(u.w, w2.w);
            }
            rm[pl][cl] = u;
        }
        __syncthreads();
    }
    if (pl == 0 && c < C) {
        const float4 t = rs[0][cl];
        const float hw = (float)HW;
        *(float4*)(stat + ((long long)n * 2 + 0) * C + c) = make_float4(t.x / hw, t.y / hw, t.z / hw, t.w / hw);
        *(float4*)(stat + ((long long)n * 2 + 1) * C + c) = rm[0][cl];
        if (ARG) *(int4*)(argmax + (long long)n * C + c) = ri[0][cl];
    }
}

// dx[n][i][c] += mask[n][i] * (dmean[n][c] / HW + [i == argmax[n][c]] * dmax[n][c]), one thread per (pixel, 4 channels)
__global__ void cm_bwd_arg_kernel(const float* __restrict__ mask, const float* __restrict__ dstat,
                                  const int* __restrict__ argmax, float* __restrict__ dx, int ldd, int N, int HW, int C4) {
    const long long total = (long long)N * HW * C4;
    const float hw = (float)HW;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4) * 4;
        const long long pix = t / C4;
        const int i = (int)(pix % HW), n = (int)(pix / HW);
        const int C = C4 * 4;
        const float k = mask[pix];
        const float4 g0 = *(const float4*)(dstat + ((long long)n * 2 + 0) * C + c);
        const float4 gx = *(const float4*)(dstat + ((long long)n * 2 + 1) * C + c);
        const int4 am = *(const int4*)(argmax + (long long)n * C + c);
        float* d = dx + pix * ldd + c;
        float4 v = *(float4*)d;
        v.x += k * (g0.x / hw + (i == am.x ? gx.x : 0.f));
        v.y += k * (g0.y / hw + (i == am.y ? gx.y : 0.f));
        v.z += k * (g0.z / hw + (i == am.z ? gx.z : 0.f));
        v.w += k * (g0.w / hw + (i == am.w ? gx.w : 0.f));
        *(float4*)d = v;
    }
}

__global__ __launch_bounds__(1024) void cm_bwd4_kernel(const float* __restrict__ x, int ldx,
                                                       const float* __restrict__ mask, const float* __restrict__ dstat,
                                                       float* __restrict__ dx, int ldd, int HW, int C) {
    __shared__ float4 rm[64][16];
    __shared__ int4 ri[64][16];
    const int n = blockIdx.y;
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const float* mk = mask + (long long)n * HW;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (c < C) {
        const float* b = x + (long long)n * HW * ldx + c;
        for (int i = pl; i < HW; i += 64) {
            const float4 v4 = *(const float4*)(b + (long long)i * ldx);
            const float k = mk[i];
            const float v[4] = {v4.x * k, v4.y * k, v4.z * k, v4.w * k};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > mx[e]) {
                    mx[e] = v[e];
                    arg[e] = i;
                }
        }
    }
    rm[pl][cl] = make_float4(mx[0], mx[1], mx[2], mx[3]);
    ri[pl][cl] = make_int4(arg[0], arg[1], arg[2], arg[3]);
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {        // first (smallest-index) maximum wins, as torch.max on the CPU
        if (pl < o) {
            float4 a = rm[pl][cl], b2 = rm[pl + o][cl];
            int4 ia = ri[pl][cl], ib = ri[pl + o][cl];
            if (b2.x > a.x || (b2.x == a.x && ib.x < ia.x)) { a.x = b2.x; ia.x = ib.x; }
            if (b2.y > a.y || (b2.y == a.y && ib.y < ia.y)) { a.y = b2.y; ia.y = ib.y; }
            if (b2.z > a.z || (b2.z == a.z && ib.z < ia.z)) { a.z = b2.z; ia.z = ib.z; }
            if (b2.w > a.w || (b2.w == a.w && ib.w < ia.w)) { a.w = b2.w; ia.w = ib.w; }
            rm[pl][cl] = a;
            ri[pl][cl] = ia;
        }
        __syncthreads();
    }
    if (c >= C) return;
    const int4 am = ri[0][cl];
    const float hw = (float)HW;
    const float4 g0 = *(const float4*)(dstat + ((long long)n * 2 + 0) * C + c);
    const float4 gx = *(const float4*)(dstat + ((long long)n * 2 + 1) * C + c);
    const float4 gm = make_float4(g0.x / hw, g0.y / hw, g0.z / hw, g0.w / hw);
    float* d = dx + (long long)n * HW * ldd + c;
    for (int i = pl; i < HW; i += 64) {
        const float k = mk[i];
        float4 t = *(float4*)(d + (long long)i * ldd);
        t.x += k * (gm.x + (i == am.x ? gx.x : 0.f));
        t.y += k * (gm.y + (i == am.y ? gx.y : 0.f));
        t.z += k * (gm.z + (i == am.z ? gx.z : 0.f));
        t.w += k * (gm.w + (i == am.w ? gx.w : 0.f));
        *(float4*)(d + (long long)i * ldd) = t;
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

extern "C" const char* pemp_last_error(void) { return g_err; }
extern "C" int pemp_abi_version(void) { return PEMP_ABI_VERSION; }

// Device memory that no XCD's L2 caches (MTYPE UC): what one XCD writes, another reads without cache write-back /
// invalidate -- the split-K convs exchange partial tiles and arrival counters through it.  Zero-filled.
extern "C" void* pemp_uncached_alloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) {
        pemp::set_error("pemp_uncached_alloc(%zu): %s", bytes, hipGetErrorString(e));
        if (p) (void)hipFree(p);
        return nullptr;
    }
    return p;
}

// One wave that does nothing for `us` microseconds (constant 100 MHz clock; bounded by an iteration count as well): two of
// these on two streams finish in `us` when the streams map to different hardware queues and in 2 x `us` when they share one --
// how the training engine checks that its side stream really runs beside the main one (pemp_spin_us).
__global__ void spin_kernel(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && n < (1 << 24)) {
        __builtin_amdgcn_s_sleep(32);
        ++n;
    }
    if (sink && n < 0) *sink = n;       // never true: keeps the loop
}

extern "C" int pemp_spin_us(int us, void* stream) {
    PEMP_REQUIRE(us > 0 && us <= 100000, "spin_us: 1 .. 100000 us");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)us * 100ull, (int*)nullptr);
    return launch_status("spin_us");
}

// Zero the arrival counters of a split-K workspace (its first 1024 bytes) on `stream`: the kernels leave them zero, a launch
// that failed or was aborted may not have.
extern "C" int pemp_splitk_reset(void* ws, void* stream) {
    PEMP_REQUIRE(ws, "splitk_reset: null workspace");
    const hipError_t e = hipMemsetAsync(ws, 0, 1024, (hipStream_t)stream);
    if (e != hipSuccess) pemp::set_error("pemp_splitk_reset: %s", hipGetErrorString(e));
    return (int)e;
}

extern "C" int pemp_uncached_free(void* p) {
    const hipError_t e = hipFree(p);
    if (e != hipSuccess) pemp::set_error("pemp_uncached_free: %s", hipGetErrorString(e));
    return (int)e;
}

extern "C" int pemp_pack_input_nhwc4_f32(const float* img, const float* prior, float* out, int N, int H, int W,
                                         void* stream) {
    PEMP_REQUIRE(img && out && N > 0 && H > 0 && W > 0, "pack_input: bad arguments");
    PEMP_REQUIRE(((uintptr_t)out & 15) == 0, "pack_input: out must be 16-byte aligned");
    long long total = (long long)N * H * W;
    hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, prior,
                       (float4*)out, H * W, total);
    return launch_status("pack_input");
}

extern "C" int pemp_maxpool2d_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int ldx, int Ho, int Wo,
                                       int ldy, int k, int s, int p, void* stream) {
    PEMP_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "maxpool: bad arguments");
    PEMP_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, "maxpool: C/ldx/ldy must be multiples of 4");
    PEMP_REQUIRE(k > 0 && s > 0 && p >= 0 && 2 * p <= k, "maxpool: bad window");
    // floor and ceil_mode output sizes (ceil: last window must start inside input or left pad)
    auto osz = [&](int in, bool ceil) {
        int num = in + 2 * p - k;
        int o = (ceil ? (num + s - 1) / s : num / s) + 1;
        if (ceil && (o - 1) * s >= in + p) --o;
        return o;
    };
    PEMP_REQUIRE((Ho == osz(H, false) || Ho == osz(H, true)) && (Wo == osz(W, false) || Wo == osz(W, true)),
                 "maxpool: Ho/Wo (%d,%d) match neither floor (%d,%d) nor ceil (%d,%d) mode", Ho, Wo, osz(H, false),
                 osz(W, false), osz(H, true), osz(W, true));
    long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W,
                       C / 4, ldx, Ho, Wo, ldy, k, s, p);
    return launch_status("maxpool");
}

extern "C" int pemp_maxpool2d_idx_nhwc_f32(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, int Ho,
                                           int Wo, int k, int s, int p, void* stream) {
    PEMP_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_idx: bad arguments");
    PEMP_REQUIRE(k > 0 && k <= 15 && s > 0 && p >= 0 && 2 * p <= k, "maxpool_idx: bad window");
    PEMP_REQUIRE(Ho > 0 && Wo > 0 && (Ho - 1) * s - p < H && (Wo - 1) * s - p < W, "maxpool_idx: bad output size");
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_idx_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, idx, N, H, W,
                       C / 4, Ho, Wo, k, s, p);
    return launch_status("maxpool_idx");
}

extern "C" int pemp_maxpool2d_idx_bwd_nhwc_f32(const uint8_t* idx, const float* dy, float* dx, int N, int H, int W, int C,
                                               int Ho, int Wo, int k, int s, int p, void* stream) {
    PEMP_REQUIRE(idx && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_idx_bwd: bad arguments");
    PEMP_REQUIRE(k > 0 && k <= 15 && s > 0 && p >= 0 && 2 * p <= k && Ho > 0 && Wo > 0, "maxpool_idx_bwd: bad window");
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_idx_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, idx, dy, dx, N,
                       H, W, C / 4, Ho, Wo, k, s, p);
    return launch_status("maxpool_idx_bwd");
}

extern "C" int pemp_global_avgpool_nhwc_f32(const float* x, float* y, int N, int HW, int C, int ldx, void* stream) {
    PEMP_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && ldx >= C, "global_avgpool: bad arguments");
    PEMP_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
                 "global_avgpool: C/ldx must be multiples of 4 and x/y 16-byte aligned");
    hipLaunchKernelGGL(gap_kernel, dim3(cdiv(C, 64), N), dim3(1024), 0, (hipStream_t)stream, x, y, HW, C, ldx);
    return launch_status("global_avgpool");
}

extern "C" int pemp_channel_affine_multi_f32(const float* x, int ldx, int M, int C, int nb, const float* const* scale,
                                             const float* const* shift, float* const* y, const int* ldy,
                                             void* stream) {
    PEMP_REQUIRE(x && scale && shift && y && ldy, "channel_affine: null pointer");
    PEMP_REQUIRE(nb >= 1 && nb <= 4, "channel_affine: nb=%d not in 1..4", nb);
    PEMP_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldx >= C, "channel_affine: bad dims");
    AffineMulti a;
    memset(&a, 0, sizeof(a));
    for (int b = 0; b < nb; ++b) {
        PEMP_REQUIRE(scale[b] && shift[b] && y[b] && ldy[b] >= C && ldy[b] % 4 == 0, "channel_affine: bad branch %d", b);
        a.scale[b] = scale[b];
        a.shift[b] = shift[b];
        a.y[b] = y[b];
        a.ldy[b] = ldy[b];
    }
    long long total = (long long)M * (C / 4);
    hipLaunchKernelGGL(affine_multi_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (long long)M, C / 4, nb, a);
    return launch_status("channel_affine");
}

extern "C" int pemp_cm_reduce_f32(const float* x, int ldx, const float* mask_in, float* mask_out, float* stat, int N,
                                  int Hm, int Wm, int Hx, int Wx, int C, int stride, void* stream) {
    return pemp_cm_reduce_arg_f32(x, ldx, mask_in, mask_out, stat, nullptr, N, Hm, Wm, Hx, Wx, C, stride, stream);
}

extern "C" int pemp_cm_reduce_arg_f32(const float* x, int ldx, const float* mask_in, float* mask_out, float* stat,
                                      int32_t* argmax, int N, int Hm, int Wm, int Hx, int Wx, int C, int stride,
                                      void* stream) {
    PEMP_REQUIRE(mask_in && mask_out && (stat || !x), "cm_reduce: null pointer");
    PEMP_REQUIRE(!argmax || (x && C % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)stat | (uintptr_t)argmax) & 15) == 0),
                 "cm_reduce: argmax needs x, C and ldx multiples of 4 and 16-byte aligned x / stat / argmax");
    PEMP_REQUIRE(N > 0 && (!x || (C > 0 && ldx >= C)) && (stride == 1 || stride == 2), "cm_reduce: bad dims");
    PEMP_REQUIRE(Hx == (Hm + 2 - 3) / stride + 1 && Wx == (Wm + 2 - 3) / stride + 1,
                 "cm_reduce: pooled mask (%d,%d)->(%d,%d) does not match feature size", Hm, Wm, Hx, Wx);
    long long total = (long long)N * Hx * Wx;
    hipLaunchKernelGGL(mask_pool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, mask_in,
                       mask_out, N, Hm, Wm, Hx, Wx, stride);
    int e = launch_status("cm_reduce/mask_pool");
    if (e || !x) return e;
    if (argmax)
        hipLaunchKernelGGL(cm_stat4_kernel<true>, dim3(cdiv(C, 64), N), dim3(1024), 0, (hipStream_t)stream, x, ldx, mask_out,
                           stat, argmax, Hx * Wx, C);
    else if (C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)stat & 15) == 0)
        hipLaunchKernelGGL(cm_stat4_kernel<false>, dim3(cdiv(C, 64), N), dim3(1024), 0, (hipStream_t)stream, x, ldx, mask_out,
                           stat, (int*)nullptr, Hx * Wx, C);
    else
        hipLaunchKernelGGL(cm_stat_kernel, dim3(cdiv(C, 64), N), dim3(256), 0, (hipStream_t)stream, x, ldx, mask_out, stat,
                           Hx * Wx, C);
    return launch_status("cm_reduce/stat");
}

extern "C" int pemp_cm_bwd_add_arg_f32(const float* mask, const float* dstat, const int32_t* argmax, float* dx, int ldd,
                                       int N, int HW, int C, void* stream) {
    PEMP_REQUIRE(mask && dstat && argmax && dx, "cm_bwd_add_arg: null pointer");
    PEMP_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 4 == 0 && ldd >= C && ldd % 4 == 0, "cm_bwd_add_arg: bad dims");
    PEMP_REQUIRE((((uintptr_t)dstat | (uintptr_t)argmax | (uintptr_t)dx) & 15) == 0, "cm_bwd_add_arg: 16-byte alignment");
    const long long total = (long long)N * HW * (C / 4);
    hipLaunchKernelGGL(cm_bwd_arg_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, mask, dstat, argmax,
                       dx, ldd, N, HW, C / 4);
    return launch_status("cm_bwd_add_arg");
}

extern "C" int pemp_cm_bwd_add_f32(const float* x, int ldx, const float* mask, const float* dstat, float* dx, int ldd,
                                   int N, int HW, int C, void* stream) {
    PEMP_REQUIRE(x && mask && dstat && dx, "cm_bwd_add: null pointer");
    PEMP_REQUIRE(N > 0 && HW > 0 && C > 0 && ldx >= C && ldd >= C, "cm_bwd_add: bad dims");
    if (C % 4 == 0 && ldx % 4 == 0 && ldd % 4 == 0 && (((uintptr_t)x | (uintptr_t)dx | (uintptr_t)dstat) & 15) == 0)
        hipLaunchKernelGGL(cm_bwd4_kernel, dim3(cdiv(C, 64), N), dim3(1024), 0, (hipStream_t)stream, x, ldx, mask, dstat, dx,
                           ldd, HW, C);
    else
        hipLaunchKernelGGL(cm_bwd_kernel, dim3(cdiv(C, 64), N), dim3(256), 0, (hipStream_t)stream, x, ldx, mask, dstat, dx,
                           ldd, HW, C);
    return launch_status("cm_bwd_add");
}


// fp32 <-> bf16 copies for the bf16 side-figure variant of the eval engine (the stem and the head stay fp32)
namespace pemp {
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long n4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = *(const float4*)(x + 4 * i);
        const float f[4] = {v.x, v.y, v.z, v.w};
        unsigned int b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int u = __builtin_bit_cast(unsigned int, f[k]);
            b[k] = (u & 0x7FFFFFFFu) > 0x7F800000u ? ((u >> 16) | 0x40u) : ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
        }
        uint2 r;
        r.x = b[0] | (b[1] << 16);
        r.y = b[2] | (b[3] << 16);
        *(uint2*)(y + 4 * i) = r;
    }
}
__global__ void bf16_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, long long n4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const uint2 r = *(const uint2*)(x + 4 * i);
        float4 o;
        o.x = __builtin_bit_cast(float, r.x << 16);
        o.y = __builtin_bit_cast(float, r.x & 0xFFFF0000u);
        o.z = __builtin_bit_cast(float, r.y << 16);
        o.w = __builtin_bit_cast(float, r.y & 0xFFFF0000u);
        *(float4*)(y + 4 * i) = o;
    }
}
}  // namespace pemp

extern "C" int pemp_convert_f32_bf16(const float* x, void* y, long long n, void* stream) {
    PEMP_REQUIRE(x && y && n > 0 && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 7) == 0, "convert_f32_bf16: n %% 4 == 0, aligned pointers");
    const long long n4 = n / 4;
    const int grid = (int)(n4 / 256 + 1 > 8192 ? 8192 : n4 / 256 + 1);
    hipLaunchKernelGGL(pemp::f32_to_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, n4);
    return pemp::launch_status("convert_f32_bf16");
}

extern "C" int pemp_convert_bf16_f32(const void* x, float* y, long long n, void* stream) {
    PEMP_REQUIRE(x && y && n > 0 && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 7) == 0, "convert_bf16_f32: n %% 4 == 0, aligned pointers");
    const long long n4 = n / 4;
    const int grid = (int)(n4 / 256 + 1 > 8192 ? 8192 : n4 / 256 + 1);
    hipLaunchKernelGGL(pemp::bf16_to_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, y, n4);
    return pemp::launch_status("convert_bf16_f32");
}
