// Shared helpers for the libpemp_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/pemp_hip.h"

namespace pemp {

void set_error(const char* fmt, ...);

#define PEMP_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            ::pemp::set_error(__VA_ARGS__);                       \
            return -1;                                            \
        }                                                         \
    } while (0)

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum 8 per-lane values over the wave with 10 shuffles instead of 48: three "keep half, send half"
// exchanges (8 -> 4 -> 2 -> 1 value per lane) followed by a butterfly over the remaining lane bits.
// Returns, in every lane l, the wave-wide sum of v[l & 7].  Fixed order -> deterministic.
__device__ __forceinline__ float wave_sum8(const float (&v)[8]) {
    const int lane = threadIdx.x & 63;
    float a[4], b[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float keep = (lane & 1) ? v[2 * k + 1] : v[2 * k];
        const float send = (lane & 1) ? v[2 * k] : v[2 * k + 1];
        a[k] = keep + __shfl_xor(send, 1, 64);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = (lane & 2) ? a[2 * k + 1] : a[2 * k];
        const float send = (lane & 2) ? a[2 * k] : a[2 * k + 1];
        b[k] = keep + __shfl_xor(send, 2, 64);
    }
    const float keep = (lane & 4) ? b[1] : b[0];
    const float send = (lane & 4) ? b[0] : b[1];
    float c = keep + __shfl_xor(send, 4, 64);
    c += __shfl_xor(c, 8, 64);
    c += __shfl_xor(c, 16, 64);
    c += __shfl_xor(c, 32, 64);
    return c;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace pemp
