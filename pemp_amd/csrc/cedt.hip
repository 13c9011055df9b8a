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

__global__ void boundary_kernel(const int64_t* __restrict__ target, uint8_t* __restrict__ bd, int H, int W) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const int64_t* t = target + (size_t)b * H * W;
    int s = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += t[yy * W + xx] == 1;
        }
    const int m = t[i] == 1;
    const int dil = min(s, 1) - m;              // dilated - mask
    const int ero = m - max(min(s - 8, 1), 0);  // mask - eroded
    bd[(size_t)b * H * W + i] = (uint8_t)((dil + ero) != 0);
}

// Vertical distance to the nearest boundary pixel in the same column (EDT_INF if the column has none).  One block per strip
// of COLS columns: the strip's boundary bytes are staged in LDS (coalesced 64-byte row segments), then every thread sweeps ITS
// column down and up out of LDS -- the two sweeps are chains of dependent min / add, but their loads do not depend on the
// chain, so they pipeline; straight from global memory (the first version) every one of the 2 H steps waited out a full
// memory latency: 216 us for 25 label images of 375 x 500, this form ~10.
__global__ __launch_bounds__(COLS) void edt_col_kernel(const uint8_t* __restrict__ bd, uint16_t* __restrict__ g, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];         // [H][cols] boundary bytes | [H][cols] uint16 distances
    const int cols = blockDim.x;
    uint16_t* down = (uint16_t*)(tile + (size_t)((H * cols + 15) & ~15));
    const int b = blockIdx.y, x0 = blockIdx.x * cols, c = threadIdx.x;
    const uint8_t* bp = bd + (size_t)b * H * W;
    const int nc = min(cols, W - x0);
#pragma unroll 8
    for (int y = 0; y < H; ++y) tile[y * cols + c] = c < nc ? bp[(size_t)y * W + x0 + c] : 0;
    __syncthreads();
    int d = EDT_INF;
#pragma unroll 8
    for (int y = 0; y < H; ++y) {
        d = tile[y * cols + c] ? 0 : min(d + 1, EDT_INF);
        down[y * cols + c] = (uint16_t)d;
    }
    uint16_t* gp = g + (size_t)b * H * W;
    d = EDT_INF;
#pragma unroll 8
    for (int y = H - 1; y >= 0; --y) {
        d = tile[y * cols + c] ? 0 : min(d + 1, EDT_INF);
        if (c < nc) gp[(size_t)y * W + x0 + c] = (uint16_t)min(d, (int)down[y * cols + c]);
    }
}

// D^2(y, x) = min_j ( g(y, j)^2 + (x - j)^2 ), exactly, in 32-bit integers.  The candidates are visited outwards from x
// (j = x, x -+ 1, x -+ 2, ...) and the scan stops as soon as (x - j)^2 alone reaches the best value so far: a pixel at
// distance d from the boundary looks at ~2 d columns instead of all W (the first version: all W, in 64-bit arithmetic).
__global__ __launch_bounds__(256) void edt_row_kernel(const uint16_t* __restrict__ g, float* __restrict__ weight, int H,
                                                      int W, double inv_sigma2) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rowmem[];
    uint32_t* g2 = (uint32_t*)rowmem;                                       // g(y, j)^2
    const int b = blockIdx.y, y = blockIdx.x;
    const uint16_t* gp = g + ((size_t)b * H + y) * W;
    for (int j = threadIdx.x; j < W; j += 256) {
        const uint32_t v = gp[j];
        g2[j] = v * v;
    }
    __syncthreads();
    constexpr uint32_t NONE = (uint32_t)EDT_INF * EDT_INF;
    for (int x = threadIdx.x; x < W; x += 256) {
        uint32_t best = g2[x];
        const int reach = max(x, W - 1 - x);
        for (int d = 1; d <= reach; ++d) {
            const uint32_t d2 = (uint32_t)d * d;
            if (d2 >= best) break;                     // every remaining candidate is at least d^2 away in x alone
            const int jl = x - d, jr = x + d;
            if (jl >= 0) best = min(best, g2[jl] + d2);
            if (jr < W) best = min(best, g2[jr] + d2);
        }
        double dist;
        if (best >= NONE) dist = sqrt((double)(y + 1) * (y + 1) + (double)x * x);   // no boundary at all
        else dist = sqrt((double)best);
        weight[((size_t)b * H + y) * W + x] = (float)(exp(-dist * inv_sigma2) + 1.0);
    }
}

}  // namespace pemp

using namespace pemp;

extern "C" size_t pemp_cedt_workspace_bytes(int B, int H, int W) {
    return (size_t)B * H * W * (sizeof(uint16_t) + 1) + 64;
}

extern "C" int pemp_cedt_weight_f32(const int64_t* target, float* weight, void* ws, size_t ws_bytes, int B, int H, int W,
                                    float sigma, void* stream) {
    PEMP_REQUIRE(target && weight && ws && B > 0 && H > 0 && W > 0 && sigma > 0.f, "cedt_weight: bad arguments");
    PEMP_REQUIRE(ws_bytes >= pemp_cedt_workspace_bytes(B, H, W), "cedt_weight: workspace too small");
    PEMP_REQUIRE(H <= 16384 && W <= 16384, "cedt_weight: label image too large (H, W <= 16384)");
    hipStream_t st = (hipStream_t)stream;
    uint16_t* g = (uint16_t*)ws;
    uint8_t* bd = (uint8_t*)(g + (size_t)B * H * W);
    hipLaunchKernelGGL(boundary_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, st, target, bd, H, W);
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
    hipLaunchKernelGGL(edt_col_kernel, dim3(cdiv(W, cols), B), dim3(cols), col_lds, st, (const uint8_t*)bd, g, H, W);
    hipLaunchKernelGGL(edt_row_kernel, dim3(H, B), dim3(256), W * sizeof(uint32_t), st, (const uint16_t*)g, weight, H, W,
                       1.0 / ((double)sigma * (double)sigma));
    return launch_status("cedt_weight");
}
