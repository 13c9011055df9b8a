// Train-time regularisers of the PEMP encoders on the device: DropBlock2D (third-party dropblock==0.3.0,
// call sites networks/pemp_stage1.py:76,79 and backbones.py:329-353) and nn.Dropout2d
// (networks/pemp_stage2.py:67,70; backbones.py:284-305).  Random numbers come from a counter-based
// Philox4x32-10 stream (key = seed, counter = element index + offset), so a mask is a pure function of
// (seed, offset, element) -- reproducible across launches, grid shapes and hipGraph replays.  The reference's
// own random stream (torch's generator) is not reproducible here: parity of the DRAWS is unpinned; the
// arithmetic applied to a given draw is checked exactly by passing the uniforms in (`uniforms` != NULL).
#include "common.h"

namespace pemp {

__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned int h0 = (unsigned int)(p0 >> 32), l0 = (unsigned int)p0;
    const unsigned int h1 = (unsigned int)(p1 >> 32), l1 = (unsigned int)p1;
    c[0] = h1 ^ c[1] ^ k0;
    c[1] = l1;
    c[2] = h0 ^ c[3] ^ k1;
    c[3] = l0;
}

// uniform in [0,1) with 24 random bits for element `idx` of stream (seed, offset)
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned long long offset, unsigned long long idx) {
    const unsigned long long ctr = idx >> 2;
    unsigned int c[4] = {(unsigned int)ctr, (unsigned int)(ctr >> 32), (unsigned int)offset, (unsigned int)(offset >> 32)};
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned int sel = (unsigned int)(idx & 3);
    const unsigned int v = sel == 0 ? c[0] : (sel == 1 ? c[1] : (sel == 2 ? c[2] : c[3]));
    return (float)(v >> 8) * (1.0f / 16777216.0f);
}

// block_mask[n,y,x] = 1 - max_{window} (u < gamma);  window = max_pool2d(k = bs, stride 1, pad = bs/2), cropped
// by one row/column at the end for even bs  =>  rows y - bs/2 .. y - bs/2 + bs - 1.
__global__ __launch_bounds__(256) void dropblock_mask_kernel(float* __restrict__ mask, int* __restrict__ count,
                                                             const float* __restrict__ uniforms, int N, int H, int W,
                                                             float gamma, int bs, unsigned long long seed,
                                                             unsigned long long offset, const unsigned long long* __restrict__ step) {
    if (step) offset += *step << 20;                 // device-side step counter: graph replays draw fresh numbers
    const int total = N * H * W;
    int kept = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int x = i % W, y = (i / W) % H, n = i / (W * H);
        const int y0 = max(y - bs / 2, 0), y1 = min(y - bs / 2 + bs, H);
        const int x0 = max(x - bs / 2, 0), x1 = min(x - bs / 2 + bs, W);
        bool drop = false;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
                const long long j = ((long long)n * H + yy) * W + xx;
                const float u = uniforms ? uniforms[j] : philox_uniform(seed, offset, (unsigned long long)j);
                drop |= u < gamma;
            }
        mask[i] = drop ? 0.f : 1.f;
        kept += drop ? 0 : 1;
    }
    __shared__ int red[256];
    red[threadIdx.x] = kept;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0]) atomicAdd(count, red[0]);        // integer: exact and order-independent
}

// y = ((x * m) * numel) / sum(m)   -- DropBlock2D.forward's two statements, in its order; also its backward.
__global__ __launch_bounds__(256) void pixel_scale_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mask,
                                                          const int* __restrict__ count, float numel,
                                                          float* __restrict__ y, int ldy, long long M, int C4) {
    const float sum = (float)*count;
    const long long total = M * C4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        const float k = mask[m];
        const float4 v = *(const float4*)(x + m * ldx + c);
        float4 o;
        o.x = __fdiv_rn(__fmul_rn(__fmul_rn(v.x, k), numel), sum);
        o.y = __fdiv_rn(__fmul_rn(__fmul_rn(v.y, k), numel), sum);
        o.z = __fdiv_rn(__fmul_rn(__fmul_rn(v.z, k), numel), sum);
        o.w = __fdiv_rn(__fmul_rn(__fmul_rn(v.w, k), numel), sum);
        *(float4*)(y + m * ldy + c) = o;
    }
}

// Dropout2d: mask[n][c] = bernoulli(1 - p) / (1 - p)
__global__ void dropout2d_mask_kernel(float* __restrict__ mask, const float* __restrict__ uniforms, int total, float p,
                                      unsigned long long seed, unsigned long long offset,
                                      const unsigned long long* __restrict__ step) {
    if (step) offset += *step << 20;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float u = uniforms ? uniforms[i] : philox_uniform(seed, offset, (unsigned long long)i);
    mask[i] = u < 1.f - p ? __fdiv_rn(1.f, 1.f - p) : 0.f;
}

__global__ __launch_bounds__(256) void channel_scale_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mask,
                                                            float* __restrict__ y, int ldy, long long M, int HW, int C4) {
    const long long total = M * C4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        const float4 k = *(const float4*)(mask + (m / HW) * (C4 * 4) + c);
        const float4 v = *(const float4*)(x + m * ldx + c);
        float4 o;
        o.x = v.x * k.x;
        o.y = v.y * k.y;
        o.z = v.z * k.z;
        o.w = v.w * k.w;
        *(float4*)(y + m * ldy + c) = o;
    }
}

static int grid_of(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace pemp

using namespace pemp;

extern "C" int pemp_dropblock_mask_f32(float* mask, int* kept_count, const float* uniforms, int N, int H, int W,
                                       float drop_prob, int block_size, uint64_t seed, uint64_t offset, const uint64_t* step,
                                       void* stream) {
    PEMP_REQUIRE(mask && kept_count, "dropblock_mask: null pointer");
    PEMP_REQUIRE(N > 0 && H > 0 && W > 0 && (long long)N * H * W < (1 << 24), "dropblock_mask: bad dims");
    PEMP_REQUIRE(block_size >= 1 && block_size <= 15 && drop_prob >= 0.f && drop_prob < 1.f, "dropblock_mask: bad parameters");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(kept_count, 0, sizeof(int), st);
    if (e != hipSuccess) {
        set_error("dropblock_mask: memset: %s", hipGetErrorString(e));
        return (int)e;
    }
    const float gamma = drop_prob / (float)(block_size * block_size);
    hipLaunchKernelGGL(dropblock_mask_kernel, dim3(grid_of((long long)N * H * W)), dim3(256), 0, st, mask, kept_count, uniforms, N,
                       H, W, gamma, block_size, (unsigned long long)seed, (unsigned long long)offset, (const unsigned long long*)step);
    return launch_status("dropblock_mask");
}

extern "C" int pemp_pixel_scale_f32(const float* x, int ldx, const float* mask, const int* kept_count, float* y, int ldy,
                                    long long M, int C, void* stream) {
    PEMP_REQUIRE(x && mask && kept_count && y, "pixel_scale: null pointer");
    PEMP_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, "pixel_scale: bad dims");
    PEMP_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pixel_scale: x/y must be 16-byte aligned");
    hipLaunchKernelGGL(pixel_scale_kernel, dim3(grid_of(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, mask, kept_count,
                       (float)M, y, ldy, M, C / 4);
    return launch_status("pixel_scale");
}

extern "C" int pemp_dropout2d_mask_f32(float* mask, const float* uniforms, int N, int C, float p, uint64_t seed,
                                       uint64_t offset, const uint64_t* step, void* stream) {
    PEMP_REQUIRE(mask && N > 0 && C > 0 && p >= 0.f && p < 1.f, "dropout2d_mask: bad arguments");
    hipLaunchKernelGGL(dropout2d_mask_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, mask, uniforms, N * C, p,
                       (unsigned long long)seed, (unsigned long long)offset, (const unsigned long long*)step);
    return launch_status("dropout2d_mask");
}

extern "C" int pemp_channel_scale_f32(const float* x, int ldx, const float* mask, float* y, int ldy, int N, int HW, int C,
                                      void* stream) {
    PEMP_REQUIRE(x && mask && y, "channel_scale: null pointer");
    PEMP_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C,
                 "channel_scale: bad dims");
    PEMP_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)mask) & 15) == 0, "channel_scale: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(channel_scale_kernel, dim3(grid_of((long long)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       mask, y, ldy, (long long)N * HW, HW, C / 4);
    return launch_status("channel_scale");
}
