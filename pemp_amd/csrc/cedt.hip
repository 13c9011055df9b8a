// Boundary-distance weight map of CELossDT on the device (reference core/losses.py:17-43 computes it
// per sample on the HOST with scipy's exact Euclidean distance transform, a GPU->CPU->GPU round trip
// inside every training step):
//
//   mask = (target == 1);   s = 3x3 box sum of mask (zero padded)
//   boundary = round( (clamp(s,0,1) - mask) + (mask - clamp(s-8,0,1)) ) != 0
//   edt(P) = Euclidean distance from P to the nearest boundary pixel (0 on the boundary)
//   weight = exp(-edt / sigma^2) + 1                       (evaluated in double, stored fp32)
//
// Exact EDT in two separable passes on integers: per column the vertical distance g to the nearest
// boundary pixel of that column, then per row  D^2(y,x) = min_j ( g(y,j)^2 + (x-j)^2 )  by direct
// minimisation out of LDS, candidates visited outwards from x with an exact early exit (no
// parabola-envelope bookkeeping).  An image WITHOUT any boundary pixel has no defined transform; scipy then returns
// sqrt((y+1)^2 + x^2) (its feature transform stays at its initial value) and so does this kernel.
#include "common.h"

namespace pemp {

constexpr int EDT_INF = 32767;        // "no boundary pixel in this column": larger than any real distance (H, W <= 16384), and
                                      // EDT_INF^2 + dx^2 still fits 32 bits
constexpr int COLS = 64;              // columns per block of the column pass (halved for very tall label images)
// exp(-d / sigma^2) + 1 rounds to 1.0f once exp(..) < 2^-24, i.e. beyond d = 16.64 sigma^2: the weight of a squared distance
// k is looked up in a table of (16.64 sigma^2 + 1)^2 entries built per call (in double, like the reference), 1.0f beyond it
constexpr int TABLE_MAX = 1 << 20;

// boundary = round((clamp(s,0,1) - mask) + (mask - clamp(s-8,0,1))) != 0 with s the zero-padded 3x3 box sum of mask.  One
// block per 254-pixel piece of a row: every thread adds its column's three rows (3 loads instead of 9), neighbours meet in LDS.
__global__ __launch_bounds__(256) void boundary_kernel(const int64_t* __restrict__ target, uint8_t* __restrict__ bd, int H, int W) {
    __shared__ int cs[256];
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * 254 - 1 + (int)threadIdx.x;
    const int64_t* t = target + (size_t)b * H * W;
    int m = 0, sum = 0;
    if (x >= 0 && x < W) {
        m = t[(size_t)y * W + x] == 1;
        sum = m;
        if (y > 0) sum += t[(size_t)(y - 1) * W + x] == 1;
        if (y + 1 < H) sum += t[(size_t)(y + 1) * W + x] == 1;
    }
    cs[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 255 || x >= W) return;
    const int s9 = cs[threadIdx.x - 1] + sum + cs[threadIdx.x + 1];
    const int dil = min(s9, 1) - m;             // dilated - mask
    const int ero = m - max(min(s9 - 8, 1), 0); // mask - eroded
    bd[((size_t)b * H + y) * W + x] = (uint8_t)((dil + ero) != 0);
}

// Vertical distance to the nearest boundary pixel in the same column (EDT_INF if the column has none).  One block per strip
// of ``cols`` columns (a power of two <= 64): the strip's boundary bytes are staged in LDS by all 256 threads; the strip is
// then cut into 256 / cols row SEGMENTS, one thread per (segment, column): a first look at the segment finds its first and
// last boundary row, the distances carried INTO a segment from above and below follow from the other segments' answers, and
// every thread sweeps its piece down and up out of LDS, 16 rows at a time (independent LDS reads, the dependent min / add
// chain in registers).  Straight from global memory, one thread per whole column (the first version), every one of the 2 H
// steps waited out a full memory latency: 216 us for 25 label images of 375 x 500.
__global__ __launch_bounds__(256) void edt_col_kernel(const uint8_t* __restrict__ bd, uint16_t* __restrict__ g, int H, int W,
                                                      int cols) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];         // [H][cols] boundary bytes | [H][cols] uint16 distances
    __shared__ int first_b[256], last_b[256];                              // [segment][column]
    uint16_t* __restrict__ down = (uint16_t*)(tile + (size_t)((H * cols + 15) & ~15));
    const int b = blockIdx.y, x0 = blockIdx.x * cols;
    const uint8_t* bp = bd + (size_t)b * H * W;
    const int nc = min(cols, W - x0);
    const int c = threadIdx.x & (cols - 1), seg = threadIdx.x / cols, nseg = 256 / cols;
    {
#pragma unroll 16
        for (int y = seg; y < H; y += nseg) tile[y * cols + c] = c < nc ? bp[(size_t)y * W + x0 + c] : 0;
    }
    __syncthreads();
    const int hs = (H + nseg - 1) / nseg, ya = min(seg * hs, H), yb = min(ya + hs, H);      // this thread's rows [ya, yb)
    int fb = H, lb = -1;
    for (int y0 = ya; y0 < yb; y0 += 16) {
        uint8_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = tile[min(y0 + k, yb - 1) * cols + c];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (v[k] && y0 + k < yb) {
                fb = min(fb, y0 + k);
                lb = y0 + k;
            }
    }
    first_b[threadIdx.x] = fb;
    last_b[threadIdx.x] = lb;
    __syncthreads();
    int above = -1, below = H;                     // nearest boundary row in the segments above / below
    for (int s2 = 0; s2 < seg; ++s2) above = max(above, last_b[s2 * cols + c]);
    for (int s2 = nseg - 1; s2 > seg; --s2) below = min(below, first_b[s2 * cols + c]);
    int d = above >= 0 ? min(ya - 1 - above, EDT_INF) : EDT_INF;       // the distance at row ya - 1
    for (int y0 = ya; y0 < yb; y0 += 16) {
        uint8_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = tile[min(y0 + k, yb - 1) * cols + c];
        uint16_t o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            d = v[k] ? 0 : min(d + 1, EDT_INF);
            o[k] = (uint16_t)d;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (y0 + k < yb) down[(y0 + k) * cols + c] = o[k];
    }
    uint16_t* gp = g + (size_t)b * H * W;
    d = below < H ? min(below - yb, EDT_INF) : EDT_INF;                 // the distance at row yb
    for (int y0 = yb - 1; y0 >= ya; y0 -= 16) {
        uint8_t v[16];
        uint16_t dn[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int y = max(y0 - k, ya);
            v[k] = tile[y * cols + c];
            dn[k] = down[y * cols + c];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (y0 - k < ya) break;
            d = v[k] ? 0 : min(d + 1, EDT_INF);
            if (c < nc) gp[(size_t)(y0 - k) * W + x0 + c] = (uint16_t)min(d, (int)dn[k]);
        }
    }
}

// weight of every squared distance below the cut-off: (float)(exp(-sqrt(k) / sigma^2) + 1), evaluated in double
__global__ void edt_table_kernel(float* __restrict__ table, int n, double inv_sigma2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) table[k] = (float)(exp(-sqrt((double)k) * inv_sigma2) + 1.0);
}

// D^2(y, x) = min_j ( g(y, j)^2 + (x - j)^2 ), exactly, in 32-bit integers.  The candidates are visited outwards from x,
// eight on either side per round (16 independent LDS reads at immediate offsets from two pointers), and the scan stops as
// soon as (x - j)^2 alone reaches the best value so far: a pixel at distance d from the boundary looks at ~2 d columns instead
// of all W (the first version: all W, in 64-bit arithmetic, and a double-precision exp + sqrt per pixel where a table lookup
// by the integer D^2 does).  The row sits in LDS between two margins of ``pad`` "no candidate" entries, so that no index needs
// a clamp: pad = W + 8 (WIDE rows, W > 4096: pad = 8 and clamped indices).
template <bool WIDE>
__global__ __launch_bounds__(256) void edt_row_kernel(const uint16_t* __restrict__ g, const float* __restrict__ table, int ntable,
                                                      float* __restrict__ weight, int H, int W, int pad) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rowmem[];
    uint32_t* g2 = (uint32_t*)rowmem;                                       // [pad | W | pad]: g(y, j)^2
    constexpr uint32_t NONE = (uint32_t)EDT_INF * EDT_INF, FAR = 0xC0000000u;
    const int b = blockIdx.y, y = blockIdx.x;
    const uint16_t* gp = g + ((size_t)b * H + y) * W;
    for (int j = threadIdx.x; j < W + 2 * pad; j += 256) {
        const int jj = j - pad;
        const uint32_t v = (jj >= 0 && jj < W) ? gp[jj] : 0;
        g2[j] = (jj >= 0 && jj < W) ? v * v : FAR;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        const uint32_t* c = g2 + pad + x;
        uint32_t best = c[0];
        const int reach = max(x, W - 1 - x);
        for (int d0 = 1; d0 <= reach; d0 += 8) {
            const uint32_t d2 = (uint32_t)d0 * d0, t = 2u * d0;
            if (d2 >= best) break;                     // every remaining candidate is at least d0^2 away in x alone
            uint32_t l[8], r[8];
            if constexpr (WIDE) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    l[k] = c[max(-(d0 + k), -pad - x)];
                    r[k] = c[min(d0 + k, W + pad - 1 - x)];
                }
            } else {
                const uint32_t* pl = c - d0;
                const uint32_t* pr = c + d0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    l[k] = pl[-k];
                    r[k] = pr[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)                 // (d0 + k)^2 = d0^2 + 2 d0 k + k^2;  FAR + d^2 < 2^32: never the minimum
                best = min(best, min(l[k], r[k]) + (d2 + (uint32_t)k * t + (uint32_t)(k * k)));
        }
        uint32_t key = best;
        if (best >= NONE) key = (uint32_t)(y + 1) * (y + 1) + (uint32_t)x * x;       // no boundary at all: scipy's sqrt((y+1)^2 + x^2)
        weight[((size_t)b * H + y) * W + x] = key < (uint32_t)ntable ? table[key] : 1.0f;
    }
}

}  // namespace pemp

using namespace pemp;

static int edt_table_len(float sigma) {
    const double cut = 16.64 * (double)sigma * (double)sigma + 2.0;       // exp(-d / sigma^2) < 2^-24 beyond: the weight is 1.0f
    const double n = cut * cut;
    return n > (double)TABLE_MAX ? TABLE_MAX : (int)n;
}

extern "C" size_t pemp_cedt_workspace_bytes(int B, int H, int W) {
    return (size_t)B * H * W * (sizeof(uint16_t) + 1) + 64 + (size_t)TABLE_MAX * sizeof(float);
}

extern "C" int pemp_cedt_weight_f32(const int64_t* target, float* weight, void* ws, size_t ws_bytes, int B, int H, int W,
                                    float sigma, void* stream) {
    PEMP_REQUIRE(target && weight && ws && B > 0 && H > 0 && W > 0 && sigma > 0.f, "cedt_weight: bad arguments");
    PEMP_REQUIRE(ws_bytes >= pemp_cedt_workspace_bytes(B, H, W), "cedt_weight: workspace too small");
    PEMP_REQUIRE(H <= 16384 && W <= 16384 && B <= 65535, "cedt_weight: label images too large (H, W <= 16384, B <= 65535)");
    const double inv_sigma2 = 1.0 / ((double)sigma * (double)sigma);
    const int ntable = edt_table_len(sigma);
    // a table that ends before the cut-off would turn real weights into 1.0f: sigma must stay below ~7.8 (the reference uses 5)
    PEMP_REQUIRE(ntable < TABLE_MAX || 16.64 * sigma * sigma + 2.0 <= 1024.0, "cedt_weight: sigma too large for the weight table");
    hipStream_t st = (hipStream_t)stream;
    float* table = (float*)ws;
    uint16_t* g = (uint16_t*)(table + TABLE_MAX);
    uint8_t* bd = (uint8_t*)(g + (size_t)B * H * W);
    hipLaunchKernelGGL(edt_table_kernel, dim3(cdiv(ntable, 256)), dim3(256), 0, st, table, ntable, inv_sigma2);
    hipLaunchKernelGGL(boundary_kernel, dim3(cdiv(W, 254), H, B), dim3(256), 0, st, target, bd, H, W);
    // column pass: H x cols boundary bytes + H x cols uint16 distances in LDS; 64 columns per block up to H = 800 (154 KB of
    // the CU's 160 KB), fewer for taller label images
    int cols = COLS;
    auto col_bytes = [&](int cc) { return (size_t)((H * cc + 15) & ~15) + (size_t)H * cc * sizeof(uint16_t); };
    while (cols > 2 && col_bytes(cols) > 156 * 1024) cols >>= 1;
    const size_t col_lds = col_bytes(cols);
    if (col_lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)edt_col_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
        if (e != hipSuccess) {
            set_error("cedt_weight: hipFuncSetAttribute(lds=%zu): %s", col_lds, hipGetErrorString(e));
            return (int)e;
        }
    }
    hipLaunchKernelGGL(edt_col_kernel, dim3(cdiv(W, cols), B), dim3(256), col_lds, st, (const uint8_t*)bd, g, H, W, cols);
    if (W <= 4096) {
        const int pad = W + 8;
        hipLaunchKernelGGL(edt_row_kernel<false>, dim3(H, B), dim3(256), (size_t)(W + 2 * pad) * sizeof(uint32_t), st,
                           (const uint16_t*)g, (const float*)table, ntable, weight, H, W, pad);
    } else {
        hipError_t e = hipFuncSetAttribute((const void*)edt_row_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((W + 16) * sizeof(uint32_t)));
        if (e != hipSuccess) {
            set_error("cedt_weight: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return (int)e;
        }
        hipLaunchKernelGGL(edt_row_kernel<true>, dim3(H, B), dim3(256), (size_t)(W + 16) * sizeof(uint32_t), st,
                           (const uint16_t*)g, (const float*)table, ntable, weight, H, W, 8);
    }
    return launch_status("cedt_weight");
}
