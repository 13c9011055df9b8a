// Episode input pipeline on the device: Pillow-exact resize / colour jitter / flip / crop / normalise of
// decoded uint8 samples (reference data_kits/pascal_voc.py:141-146,185-237; arithmetic of Pillow's
// Resample.c, Geometry.c, Blend.c, Convert.c restated -- see oracle/pil_ops.py for the CPU checker).
// Byte/integer work, HBM- and launch-bound: one launch per stage for the whole batch of samples
// (blockIdx.y = sample), descriptors and pixels arrive in one uploaded blob.
#include <math.h>
#include <string.h>
#include "common.h"

namespace pemp {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Resample.c, 8 bits per channel
constexpr int MAX_TAPS = 64;

struct SampleLayout {
    long long xb, xk, yb, yk, nx, ny, tmp, res, gsum, end;
};

__host__ __device__ inline long long align16(long long v) { return (v + 15) & ~15LL; }

// Byte offsets (relative to desc.ws_off) of one sample's tables and intermediates.
__host__ __device__ inline SampleLayout sample_layout(const pemp_sample_desc& d) {
    SampleLayout L;
    long long o = 0;
    L.xb = o; o = align16(o + (long long)d.sw * 2 * 4);
    L.xk = o; o = align16(o + (long long)d.sw * d.ksx * 4);
    L.yb = o; o = align16(o + (long long)d.sh * 2 * 4);
    L.yk = o; o = align16(o + (long long)d.sh * d.ksy * 4);
    L.nx = o; o = align16(o + (long long)d.sw * 4);
    L.ny = o; o = align16(o + (long long)d.sh * 4);
    L.gsum = o; o = align16(o + 8);
    L.tmp = o; o = align16(o + (d.img_off >= 0 ? (long long)d.hs * d.sw * 3 : 0));
    L.res = o; o = align16(o + (d.img_off >= 0 ? (long long)d.sh * d.sw * 3 : 0));
    L.end = o;
    return L;
}

static int taps_for(int in_size, int out_size) {
    double fs = (double)in_size / out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(fs) * 2 + 1;
}

// ---------------------------------------------------------------------------------------------
// precompute_coeffs + normalize_coeffs_8bpc (bilinear, full box) and the ImagingScaleAffine index table.
// Every double operation is an explicit correctly-rounded intrinsic: no FMA contraction, so the tables are the
// ones the host library computes.
__global__ __launch_bounds__(256) void coeffs_kernel(const pemp_sample_desc* __restrict__ descs, char* __restrict__ ws) {
    const pemp_sample_desc d = descs[blockIdx.y];
    const SampleLayout L = sample_layout(d);
    char* base = ws + d.ws_off;
    const int axis = blockIdx.z;
    const int in_size = axis ? d.hs : d.ws, out_size = axis ? d.sh : d.sw, ks = axis ? d.ksy : d.ksx;
    int* bounds = (int*)(base + (axis ? L.yb : L.xb));
    int* kk = (int*)(base + (axis ? L.yk : L.xk));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (axis == 0) *(unsigned long long*)(base + L.gsum) = 0ull;
        if (d.msk_off >= 0 && (d.mask_mode == 1 || d.mask_mode == 2)) {
            int* tab = (int*)(base + (axis ? L.ny : L.nx));
            const double a = __ddiv_rn((double)in_size, (double)out_size);
            double xo = __dmul_rn(a, 0.5);
            for (int i = 0; i < out_size; ++i) {
                int v = (int)xo;
                tab[i] = v < in_size ? v : in_size - 1;
                xo = __dadd_rn(xo, a);
            }
        }
    }
    if (d.img_off < 0) return;
    const int xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = filterscale;                     // bilinear support 1.0 * filterscale
    const double ss = __ddiv_rn(1.0, filterscale);
    const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
    int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > ks) xmax = ks;                                // cannot happen (ks = ceil(support)*2+1); keeps stores in range
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
        t = fabs(t);
        ww = __dadd_rn(ww, t < 1.0 ? __dsub_rn(1.0, t) : 0.0);
    }
    for (int x = 0; x < ks; ++x) {
        int q = 0;
        if (x < xmax) {
            double t = fabs(__dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss));
            double w = t < 1.0 ? __dsub_rn(1.0, t) : 0.0;
            if (ww != 0.0) w = __ddiv_rn(w, ww);
            const double sc = __dmul_rn(w, (double)(1 << PRECISION_BITS));
            q = w < 0 ? (int)__dadd_rn(-0.5, sc) : (int)__dadd_rn(0.5, sc);
        }
        kk[(long long)xx * ks + x] = q;
    }
    bounds[xx * 2] = xmin;
    bounds[xx * 2 + 1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// ImagingResampleHorizontal_8bpc: tmp[y][x][c] over the source rows.
__global__ __launch_bounds__(256) void hpass_kernel(const unsigned char* __restrict__ blob, const pemp_sample_desc* __restrict__ descs,
                                                    char* __restrict__ ws) {
    const pemp_sample_desc d = descs[blockIdx.y];
    if (d.img_off < 0) return;
    const long long idx = blockIdx.x * 256LL + threadIdx.x;
    if (idx >= (long long)d.hs * d.sw) return;
    const SampleLayout L = sample_layout(d);
    char* base = ws + d.ws_off;
    const int y = (int)(idx / d.sw), x = (int)(idx - (long long)y * d.sw);
    const int* bounds = (const int*)(base + L.xb);
    const int* k = (const int*)(base + L.xk) + (long long)x * d.ksx;
    const int x0 = bounds[x * 2], n = bounds[x * 2 + 1];
    const unsigned char* src = blob + d.img_off + ((long long)y * d.ws + x0) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int c = k[i];
        s0 += src[i * 3 + 0] * c;
        s1 += src[i * 3 + 1] * c;
        s2 += src[i * 3 + 2] * c;
    }
    unsigned char* o = (unsigned char*)(base + L.tmp) + idx * 3;
    o[0] = clip8(s0);
    o[1] = clip8(s1);
    o[2] = clip8(s2);
}

// ImagingResampleVertical_8bpc: res[y][x][c] from tmp.
__global__ __launch_bounds__(256) void vpass_kernel(const pemp_sample_desc* __restrict__ descs, char* __restrict__ ws) {
    const pemp_sample_desc d = descs[blockIdx.y];
    if (d.img_off < 0) return;
    const long long idx = blockIdx.x * 256LL + threadIdx.x;
    if (idx >= (long long)d.sh * d.sw) return;
    const SampleLayout L = sample_layout(d);
    char* base = ws + d.ws_off;
    const int y = (int)(idx / d.sw), x = (int)(idx - (long long)y * d.sw);
    const int* bounds = (const int*)(base + L.yb);
    const int* k = (const int*)(base + L.yk) + (long long)y * d.ksy;
    const int y0 = bounds[y * 2], n = bounds[y * 2 + 1];
    const unsigned char* src = (const unsigned char*)(base + L.tmp) + ((long long)y0 * d.sw + x) * 3;
    const long long row = (long long)d.sw * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int c = k[i];
        s0 += src[i * row + 0] * c;
        s1 += src[i * row + 1] * c;
        s2 += src[i * row + 2] * c;
    }
    unsigned char* o = (unsigned char*)(base + L.res) + idx * 3;
    o[0] = clip8(s0);
    o[1] = clip8(s1);
    o[2] = clip8(s2);
}

// ---------------------------------------------------------------------------------------------
// ColorJitter stage = one ImageEnhance op per sample (op id from the descriptor), in place on `res`.
__device__ __forceinline__ int rgb2l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__global__ __launch_bounds__(256) void gray_sum_kernel(const pemp_sample_desc* __restrict__ descs, char* __restrict__ ws, int stage) {
    const pemp_sample_desc d = descs[blockIdx.y];
    if (d.img_off < 0 || ((d.jitter_order >> (2 * stage)) & 3) != 2) return;
    const SampleLayout L = sample_layout(d);
    char* base = ws + d.ws_off;
    const long long total = (long long)d.sh * d.sw;
    const unsigned char* p = (const unsigned char*)(base + L.res);
    unsigned int s = 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        s += (unsigned int)rgb2l(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
    __shared__ unsigned int red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0]) atomicAdd((unsigned long long*)(base + L.gsum), (unsigned long long)red[0]);   // integer: exact
}

__device__ __forceinline__ unsigned char blend8(float deg, float v, float a, bool interp) {
    const float t = __fadd_rn(deg, __fmul_rn(a, __fsub_rn(v, deg)));          // Blend.c: in1 + alpha * (in2 - in1), float
    if (interp) return (unsigned char)(int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)(int)t);
}

// 4 pixels (12 bytes = three aligned dwords of the flat byte image) per thread.
__global__ __launch_bounds__(256) void jitter_kernel(const pemp_sample_desc* __restrict__ descs, char* __restrict__ ws, int stage) {
    const pemp_sample_desc* dp = descs + blockIdx.y;
    const int op = (dp->jitter_order >> (2 * stage)) & 3;
    if (dp->img_off < 0 || op == 0) return;
    const long long total = (long long)dp->sh * dp->sw;
    const long long px0 = (blockIdx.x * 256LL + threadIdx.x) * 4;
    if (px0 >= total) return;
    const pemp_sample_desc d = *dp;
    const SampleLayout L = sample_layout(d);
    char* base = ws + d.ws_off;
    const float a = op == 1 ? d.jitter[0] : (op == 2 ? d.jitter[1] : d.jitter[2]);
    const bool interp = a >= 0.f && a <= 1.f;
    float cmean = 0.f;
    if (op == 2) {                                                            // Contrast: int(mean(L) + 0.5) everywhere
        const unsigned long long s = *(const unsigned long long*)(base + L.gsum);
        cmean = (float)(int)__dadd_rn(__ddiv_rn((double)s, (double)total), 0.5);
    }
    unsigned char* p = (unsigned char*)(base + L.res) + px0 * 3;
    unsigned char v[12];
    const int npx = total - px0 >= 4 ? 4 : (int)(total - px0);
    if (npx == 4) {
        const unsigned int* q = (const unsigned int*)p;
        *(unsigned int*)(v + 0) = q[0];
        *(unsigned int*)(v + 4) = q[1];
        *(unsigned int*)(v + 8) = q[2];
    } else {
        for (int i = 0; i < npx * 3; ++i) v[i] = p[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = v[i * 3], g = v[i * 3 + 1], b = v[i * 3 + 2];
        const float deg = op == 1 ? 0.f : (op == 2 ? cmean : (float)rgb2l(r, g, b));   // black / mean gray / gray image
        v[i * 3 + 0] = blend8(deg, (float)r, a, interp);
        v[i * 3 + 1] = blend8(deg, (float)g, a, interp);
        v[i * 3 + 2] = blend8(deg, (float)b, a, interp);
    }
    if (npx == 4) {
        unsigned int* q = (unsigned int*)p;
        q[0] = *(unsigned int*)(v + 0);
        q[1] = *(unsigned int*)(v + 4);
        q[2] = *(unsigned int*)(v + 8);
    } else {
        for (int i = 0; i < npx * 3; ++i) p[i] = v[i];
    }
}

// ---------------------------------------------------------------------------------------------
// flip + crop window + ToTensor + Normalize -> fp32 [3][H][W]
struct Norm3 {
    float mean[3], std[3];
};

__global__ __launch_bounds__(256) void finish_kernel(const pemp_sample_desc* __restrict__ descs, const char* __restrict__ ws,
                                                     float* __restrict__ out, int H, int W, Norm3 nm) {
    const pemp_sample_desc d = descs[blockIdx.y];
    if (d.img_off < 0) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const SampleLayout L = sample_layout(d);
    const int y = idx / W, x = idx - y * W;
    const int sy = d.oy + y, sx = d.flip ? d.sw - 1 - (d.ox + x) : d.ox + x;
    const unsigned char* p = (const unsigned char*)(ws + d.ws_off + L.res) + ((long long)sy * d.sw + sx) * 3;
    float* o = out + d.img_out + idx;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        o[(long long)c * H * W] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p[c], 255.f), nm.mean[c]), nm.std[c]);
}

// label path: nearest resize (+flip, crop) -> support planes / int64 label; or source-size int64 label.
__global__ __launch_bounds__(256) void label_kernel(const unsigned char* __restrict__ blob, const pemp_sample_desc* __restrict__ descs,
                                                    const char* __restrict__ ws, float* __restrict__ planes,
                                                    long long* __restrict__ labels, int H, int W) {
    const pemp_sample_desc d = descs[blockIdx.y];
    if (d.msk_off < 0 || d.mask_mode == 0) return;
    const unsigned char* src = blob + d.msk_off;
    const long long idx = blockIdx.x * 256LL + threadIdx.x;
    if (d.mask_mode == 3) {
        if (idx < (long long)d.hs * d.ws) labels[d.msk_out + idx] = src[idx] / 255;
        return;
    }
    if (idx >= (long long)H * W) return;
    const SampleLayout L = sample_layout(d);
    const int* nx = (const int*)(ws + d.ws_off + L.nx);
    const int* ny = (const int*)(ws + d.ws_off + L.ny);
    const int y = (int)(idx / W), x = (int)(idx - (long long)y * W);
    const int sy = ny[d.oy + y], sx = nx[d.flip ? d.sw - 1 - (d.ox + x) : d.ox + x];
    const int v = src[(long long)sy * d.ws + sx] / 255;
    if (d.mask_mode == 1) {
        planes[d.msk_out + idx] = (float)v;
        planes[d.msk_out + (long long)H * W + idx] = 1.f - (float)v;
    } else {
        labels[d.msk_out + idx] = v;
    }
}

static bool desc_ok(const pemp_sample_desc& d, int i, int H, int W) {
    const int lim = 16384;
    if (d.img_off < 0 && d.msk_off < 0) { set_error("episode: sample %d has neither image nor label", i); return false; }
    if (d.hs <= 0 || d.ws <= 0 || d.hs > lim || d.ws > lim) { set_error("episode: sample %d source size %dx%d", i, d.hs, d.ws); return false; }
    const bool resized = d.img_off >= 0 || d.mask_mode == 1 || d.mask_mode == 2;
    if (resized) {
        if (d.sh <= 0 || d.sw <= 0 || d.sh > lim || d.sw > lim) { set_error("episode: sample %d resized size %dx%d", i, d.sh, d.sw); return false; }
        if (d.oy < 0 || d.ox < 0 || d.oy + H > d.sh || d.ox + W > d.sw) {
            set_error("episode: sample %d crop window (%d,%d)+(%d,%d) outside %dx%d", i, d.oy, d.ox, H, W, d.sh, d.sw);
            return false;
        }
    }
    if (d.mask_mode < 0 || d.mask_mode > 3 || (d.mask_mode != 0 && d.msk_off < 0)) { set_error("episode: sample %d mask_mode %d", i, d.mask_mode); return false; }
    if (d.jitter_order < 0 || d.jitter_order > 63) { set_error("episode: sample %d jitter_order %d", i, d.jitter_order); return false; }
    int n_contrast = 0;
    for (int s = 0; s < 3; ++s) n_contrast += ((d.jitter_order >> (2 * s)) & 3) == 2;
    if (n_contrast > 1) { set_error("episode: sample %d applies contrast twice", i); return false; }
    return true;
}

}  // namespace pemp

using namespace pemp;

extern "C" size_t pemp_episode_plan(pemp_sample_desc* descs, int n, int H, int W) {
    if (!descs || n <= 0 || H <= 0 || W <= 0) {
        set_error("episode_plan: bad arguments");
        return 0;
    }
    long long off = 0;
    for (int i = 0; i < n; ++i) {
        pemp_sample_desc& d = descs[i];
        if (!desc_ok(d, i, H, W)) return 0;
        if (d.sh <= 0 || d.sw <= 0) d.sh = d.hs, d.sw = d.ws;          // mode-3-only samples: tables unused
        d.ksx = taps_for(d.ws, d.sw);
        d.ksy = taps_for(d.hs, d.sh);
        if (d.ksx > MAX_TAPS || d.ksy > MAX_TAPS) {
            set_error("episode_plan: sample %d shrinks by more than %dx", i, (MAX_TAPS - 1) / 2);
            return 0;
        }
        d.ws_off = off;
        off += sample_layout(d).end;
    }
    return (size_t)(off + 16);
}

extern "C" int pemp_episode_preprocess(const uint8_t* blob, const pemp_sample_desc* dh, const pemp_sample_desc* dd, int n,
                                       int H, int W, const float* mean, const float* std, float* img_out,
                                       float* planes_out, int64_t* label_out, void* ws, size_t ws_bytes, void* stream) {
    PEMP_REQUIRE(blob && dh && dd && mean && std && ws, "episode_preprocess: null pointer");
    PEMP_REQUIRE(n > 0 && n <= 65535 && H > 0 && W > 0, "episode_preprocess: bad dims");
    PEMP_REQUIRE(((uintptr_t)ws & 15) == 0, "episode_preprocess: workspace must be 16-byte aligned");
    long long max_out = 0, max_h = 0, max_v = 0, max_lab = 0, need = 0;
    bool any_img = false, any_lab = false;
    int stages = 0;
    for (int i = 0; i < n; ++i) {
        const pemp_sample_desc& d = dh[i];
        if (!desc_ok(d, i, H, W)) return -1;
        PEMP_REQUIRE(d.ksx == taps_for(d.ws, d.sw) && d.ksy == taps_for(d.hs, d.sh) && d.ws_off >= 0,
                     "episode_preprocess: sample %d was not planned (pemp_episode_plan)", i);
        need = d.ws_off + sample_layout(d).end > need ? d.ws_off + sample_layout(d).end : need;
        if (d.img_off >= 0) {
            PEMP_REQUIRE(img_out, "episode_preprocess: img_out is null");
            any_img = true;
            max_out = max_out > (d.sw > d.sh ? d.sw : d.sh) ? max_out : (d.sw > d.sh ? d.sw : d.sh);
            max_h = max_h > (long long)d.hs * d.sw ? max_h : (long long)d.hs * d.sw;
            max_v = max_v > (long long)d.sh * d.sw ? max_v : (long long)d.sh * d.sw;
            for (int s = 0; s < 3; ++s)
                if ((d.jitter_order >> (2 * s)) & 3) stages = stages > s + 1 ? stages : s + 1;
        }
        if (d.mask_mode) {
            PEMP_REQUIRE(d.mask_mode == 1 ? planes_out != nullptr : label_out != nullptr, "episode_preprocess: mask output is null");
            any_lab = true;
            long long m = d.mask_mode == 3 ? (long long)d.hs * d.ws : (long long)H * W;
            max_lab = max_lab > m ? max_lab : m;
            if (d.mask_mode != 3) max_out = max_out > (d.sw > d.sh ? d.sw : d.sh) ? max_out : (d.sw > d.sh ? d.sw : d.sh);
        }
    }
    PEMP_REQUIRE((size_t)need <= ws_bytes, "episode_preprocess: workspace too small (%lld > %zu)", need, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    char* wsc = (char*)ws;
    if (max_out > 0) {
        hipLaunchKernelGGL(coeffs_kernel, dim3((unsigned)cdiv((int)max_out, 256), n, 2), dim3(256), 0, st, dd, wsc);
    }
    if (any_img) {
        hipLaunchKernelGGL(hpass_kernel, dim3((unsigned)((max_h + 255) / 256), n), dim3(256), 0, st, blob, dd, wsc);
        hipLaunchKernelGGL(vpass_kernel, dim3((unsigned)((max_v + 255) / 256), n), dim3(256), 0, st, dd, wsc);
        for (int s = 0; s < stages; ++s) {
            hipLaunchKernelGGL(gray_sum_kernel, dim3(64, n), dim3(256), 0, st, dd, wsc, s);
            hipLaunchKernelGGL(jitter_kernel, dim3((unsigned)((max_v + 1023) / 1024), n), dim3(256), 0, st, dd, wsc, s);
        }
        Norm3 nm;
        for (int c = 0; c < 3; ++c) nm.mean[c] = mean[c], nm.std[c] = std[c];
        hipLaunchKernelGGL(finish_kernel, dim3((unsigned)cdiv(H * W, 256), n), dim3(256), 0, st, dd, (const char*)wsc, img_out, H, W, nm);
    }
    if (any_lab) {
        hipLaunchKernelGGL(label_kernel, dim3((unsigned)((max_lab + 255) / 256), n), dim3(256), 0, st, blob, dd, (const char*)wsc,
                           planes_out, (long long*)label_out, H, W);
    }
    return launch_status("episode_preprocess");
}
