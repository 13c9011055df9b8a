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
// minimisation out of LDS (W <= ~640, so W^2 per row is cheap and exact; no parabola-envelope
// bookkeeping).  An image WITHOUT any boundary pixel has no defined transform; scipy then returns
// sqrt((y+1)^2 + x^2) (its feature transform stays at its initial value) and so does this kernel.
#include "common.h"

namespace pemp {

constexpr int EDT_INF = 1 << 20;

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

// vertical distance to the nearest boundary pixel in the same column (EDT_INF if the column has none)
__global__ void edt_col_kernel(const uint8_t* __restrict__ bd, int* __restrict__ g, int H, int W) {
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const uint8_t* bp = bd + (size_t)b * H * W;
    int* gp = g + (size_t)b * H * W;
    int d = EDT_INF;
    for (int y = 0; y < H; ++y) {
        d = bp[y * W + x] ? 0 : min(d + 1, EDT_INF);
        gp[y * W + x] = d;
    }
    d = EDT_INF;
    for (int y = H - 1; y >= 0; --y) {
        d = bp[y * W + x] ? 0 : min(d + 1, EDT_INF);
        gp[y * W + x] = min(gp[y * W + x], d);
    }
}

__global__ __launch_bounds__(256) void edt_row_kernel(const int* __restrict__ g, float* __restrict__ weight, int H,
                                                      int W, double inv_sigma2) {
    extern __shared__ int grow[];
    const int b = blockIdx.y, y = blockIdx.x;
    const int* gp = g + ((size_t)b * H + y) * W;
    for (int j = threadIdx.x; j < W; j += 256) grow[j] = gp[j];
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        long long best = (long long)EDT_INF * EDT_INF;
        for (int j = 0; j < W; ++j) {
            const long long gj = grow[j], dx = x - j;
            const long long v = gj * gj + dx * dx;
            best = v < best ? v : best;
        }
        double dist;
        if (best >= (long long)EDT_INF * EDT_INF) dist = sqrt((double)(y + 1) * (y + 1) + (double)x * x);   // no boundary at all
        else dist = sqrt((double)best);
        weight[((size_t)b * H + y) * W + x] = (float)(exp(-dist * inv_sigma2) + 1.0);
    }
}

}  // namespace pemp

using namespace pemp;

extern "C" size_t pemp_cedt_workspace_bytes(int B, int H, int W) {
    return (size_t)B * H * W * (sizeof(int) + 1) + 64;
}

extern "C" int pemp_cedt_weight_f32(const int64_t* target, float* weight, void* ws, size_t ws_bytes, int B, int H, int W,
                                    float sigma, void* stream) {
    PEMP_REQUIRE(target && weight && ws && B > 0 && H > 0 && W > 0 && sigma > 0.f, "cedt_weight: bad arguments");
    PEMP_REQUIRE(ws_bytes >= pemp_cedt_workspace_bytes(B, H, W), "cedt_weight: workspace too small");
    PEMP_REQUIRE(H < EDT_INF / 2 && W < 16384, "cedt_weight: image too large");
    hipStream_t st = (hipStream_t)stream;
    int* g = (int*)ws;
    uint8_t* bd = (uint8_t*)(g + (size_t)B * H * W);
    hipLaunchKernelGGL(boundary_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, st, target, bd, H, W);
    hipLaunchKernelGGL(edt_col_kernel, dim3(cdiv(W, 64), B), dim3(64), 0, st, (const uint8_t*)bd, g, H, W);
    hipLaunchKernelGGL(edt_row_kernel, dim3(H, B), dim3(256), W * sizeof(int), st, (const int*)g, weight, H, W,
                       1.0 / ((double)sigma * (double)sigma));
    return launch_status("cedt_weight");
}
