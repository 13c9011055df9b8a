// Shared between the conv engine's translation units.
#pragma once
#include "common.h"

namespace pemp {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float v4f;   // first-class vector: loads/stores never become memcpy

struct ConvArgs {
    const float* x;
    const float* w;
    float* y;
    const float* scale;
    const float* shift;
    const float* res;
    int N, H, W, Cin, ldx, Ho, Wo, Cout, ldy, KH, KW, stride, pad, dil, ldr, Kpad;
    unsigned flags;
    int M, HoWo, cin_steps, nk, ntaps;
};

// XCD-aware tile order.  Hardware deals consecutive block ids round-robin over the 8 XCDs (each
// with a private 4 MiB L2).  Tiles that share an A row-panel (same M tile, different N tiles) and
// neighbouring M tiles (3x3 halos) should therefore get ids that are congruent mod 8.  This
// bijection hands every XCD one contiguous range of the logical (M-major, N-minor) tile order.
// Placement only affects speed, never results.
__device__ __forceinline__ int xcd_tile_order(int bid, int nblk) {
    const int x = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, rem = nblk & 7;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + idx;
}

// conv_dma.hip
int launch_conv_dma(int tile, const ConvArgs& a, hipStream_t st);

}  // namespace pemp
