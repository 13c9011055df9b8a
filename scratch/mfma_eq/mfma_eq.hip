// Are v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 (and 4x4x1) bit-identical accumulators for the same K order?
// D[i][j] = sum_k A[i][k] B[k][j], K = 256, random fp32 with wide exponent spread; compares the three shapes against each other and
// against a sequential fmaf chain / a (mul, add) chain on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int K = 256;
// A: [32][K] row-major, B: [K][32] row-major, out32: [32][32]
__global__ void k32(const float* A, const float* B, float* D) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    // acc[e]: row = (e & 3) + 8 * (e >> 2) + 4 * h, col = r
    for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}
__global__ void k16(const float* A, const float* B, float* D) {      // the top-left 16 x 16 block
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 acc = {0};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + g], B[(k + g) * 32 + r], acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) D[(4 * g + e) * 32 + r] = acc[e];
}
int main() {
    std::vector<float> A(32 * K), B(K * 32), D32(1024), D16(1024, 0.f);
    srand(1);
    auto rnd = [] { float m = (rand() / (float)RAND_MAX) * 2.f - 1.f; int e = rand() % 12 - 6; return ldexpf(m, e); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    float *dA, *dB, *d32, *d16;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d32, 4096); hipMalloc(&d16, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d16, 0, 4096);
    k32<<<1, 64>>>(dA, dB, d32); k16<<<1, 64>>>(dA, dB, d16);
    hipMemcpy(D32.data(), d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(D16.data(), d16, 4096, hipMemcpyDeviceToHost);
    int same = 0, fma_same32 = 0, fma_same16 = 0, ma_same32 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float f = 0.f, m = 0.f;
        for (int k = 0; k < K; ++k) { f = fmaf(A[i * K + k], B[k * 32 + j], f); m = m + A[i * K + k] * B[k * 32 + j]; }
        same += D32[i * 32 + j] == D16[i * 32 + j];
        fma_same32 += D32[i * 32 + j] == f; fma_same16 += D16[i * 32 + j] == f; ma_same32 += D32[i * 32 + j] == m;
    }
    printf("16x16 block: 32x32x2 == 16x16x4 in %d of 256; 32x32x2 == sequential fmaf chain %d; 16x16x4 == fmaf chain %d; 32x32x2 == mul+add chain %d\n",
           same, fma_same32, fma_same16, ma_same32);
    return 0;
}
