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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace pemp
