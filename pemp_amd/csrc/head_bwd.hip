// Backward of the prototype head (training): cross-entropy + bilinear upsample, cosine map with
// group max, and the meta-prototype module (soft assignment + masked pooling) -- the reverse of
// networks/pemp_stage1.py:142-163,195-261 under F.cross_entropy(ignore_index=255)
// (core/losses.py:10).  Streaming kernels, wave per feature pixel, fixed-order partial sums.
//
// Notation (one episode b, shot s, pixel i, prototype j in group g(j); J = 2p):
//   d_ij = -|x_i - ctr_j|^2, sig_ij = softmax_{j in g}(d_i.), a_ij = sig_ij * m_g(i)
//   N_sj = sum_i a_ij x_i,  D_sj = sum_i a_ij + eps,  P_sj = N_sj / D_sj,  P_j = mean_s P_sj
//   L_ij = k * cos(y_i, P_j),  pred_g(i) = max_{j in g} L_ij,  logits = bilinear(pred)
#include "head_common.h"

namespace pemp {

// -----------------------------------------------------------------------------------------------
// per-shot prototypes from the forward workspace:  Pps[bs][j][c] = N/D,  Dps[bs][j] = D
__global__ __launch_bounds__(256) void pool_shot_kernel(const float* __restrict__ part, const float* __restrict__ asum,
                                                        const float* __restrict__ den_override,
                                                        float* __restrict__ Pps, float* __restrict__ Dps, int c, int J,
                                                        int nchunks, float eps) {
    __shared__ float red[4][64];
    const int bs = blockIdx.y, j = blockIdx.x;
    const int ch = blockIdx.z * 64 + (threadIdx.x & 63), chl = min(ch, c - 1);
    const float num = chunk_sum(part + ((size_t)bs * nchunks * J + j) * c + chl, (size_t)J * c, nchunks, red);
    float den = chunk_sum(asum + (size_t)bs * nchunks * J + j, (size_t)J, nchunks, red);
    if (den_override) den = den_override[bs * J + j];      // Baseline: exact full-resolution mask sums
    den += eps;
    if (threadIdx.x < 64 && ch < c) Pps[((size_t)bs * J + j) * c + ch] = num / den;
    if (blockIdx.z == 0 && threadIdx.x == 0) Dps[bs * J + j] = den;
}

// -----------------------------------------------------------------------------------------------
// dpred[b][ch][i] = sum_P W[P][i] * dlogits[b][ch][P]: adjoint of F.interpolate(pred, (Ho,Wo), "bilinear",
// align_corners=True).  One wave per low-resolution pixel: its 64 lanes share the full-resolution window whose
// bilinear stencils can touch the pixel (lane-strided, then a fixed butterfly -> deterministic).
//   DERIVE: dlogits = (softmax - onehot) * weight / n_valid_total, re-evaluated from `pred` (CE path)
//   else  : dlogits read from memory (autograd bridge)
template <bool DERIVE>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ pred,
                                                           const int64_t* __restrict__ target,
                                                           const float* __restrict__ weight,
                                                           const double* __restrict__ stats, int B,
                                                           const float* __restrict__ dlogits, float* __restrict__ dpred,
                                                           int h, int w, int Ho, int Wo) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = h * w;
    if (i >= n) return;
    float inv = 0.f;
    if (DERIVE) {
        double nv = 0.0;
        for (int k = 0; k < B; ++k) nv += stats[k * 8 + 1];
        inv = nv > 0.0 ? (float)(1.0 / nv) : 0.f;
    }
    const int y = i / w, x = i - y * w;
    const float* p0 = pred + (size_t)b * 2 * n;
    const float* p1 = p0 + n;
    const float* d0 = dlogits + (size_t)b * 2 * Ho * Wo;
    const float* d1 = d0 + (size_t)Ho * Wo;
    const float sh = (Ho > 1 && h > 1) ? (float)(h - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1 && w > 1) ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    const int Y0 = sh > 0.f ? max(0, (int)floorf((float)(y - 1) / sh) - 1) : 0;
    const int Y1 = sh > 0.f ? min(Ho - 1, (int)ceilf((float)(y + 1) / sh) + 1) : Ho - 1;
    const int X0 = sw > 0.f ? max(0, (int)floorf((float)(x - 1) / sw) - 1) : 0;
    const int X1 = sw > 0.f ? min(Wo - 1, (int)ceilf((float)(x + 1) / sw) + 1) : Wo - 1;
    const int nx = X1 - X0 + 1, npts = (Y1 - Y0 + 1) * nx;
    float g0 = 0.f, g1 = 0.f;
    for (int t = lane; t < npts; t += 64) {
        const int Y = Y0 + t / nx, X = X0 + t % nx;
        const Bilin by = bilin(Y, h, Ho);
        const float wy = (by.i0 == y ? 1.f - by.l : 0.f) + (by.i1 == y && by.i1 != by.i0 ? by.l : 0.f) +
                         (by.i1 == by.i0 && by.i0 == y ? by.l : 0.f);
        const Bilin bx = bilin(X, w, Wo);
        const float wx = (bx.i0 == x ? 1.f - bx.l : 0.f) + (bx.i1 == x && bx.i1 != bx.i0 ? bx.l : 0.f) +
                         (bx.i1 == bx.i0 && bx.i0 == x ? bx.l : 0.f);
        if (wy == 0.f || wx == 0.f) continue;
        if (DERIVE) {
            const int tg = (int)target[((size_t)b * Ho + Y) * Wo + X];
            if (tg == 255) continue;
            const float l0 = bilerp(p0, w, by, bx), l1 = bilerp(p1, w, by, bx);
            const float m = fmaxf(l0, l1);
            const float e0 = expf(l0 - m), e1 = expf(l1 - m);
            const float s = e0 + e1;
            const float wgt = wy * wx * inv * (weight ? weight[((size_t)b * Ho + Y) * Wo + X] : 1.f);
            g0 += wgt * (e0 / s - (tg == 0 ? 1.f : 0.f));
            g1 += wgt * (e1 / s - (tg == 1 ? 1.f : 0.f));
        } else {
            g0 += wy * wx * d0[(size_t)Y * Wo + X];
            g1 += wy * wx * d1[(size_t)Y * Wo + X];
        }
    }
    g0 = wave_sum(g0);
    g1 = wave_sum(g1);
    if (lane == 0) {
        dpred[((size_t)b * 2 + 0) * n + i] = g0;
        dpred[((size_t)b * 2 + 1) * n + i] = g1;
    }
}

// -----------------------------------------------------------------------------------------------
// cosine + group-max backward.  One wave per query pixel.
//   dY_i   = sum_g k*gp_g(i) * (v_j - cos_ij u_i) / |y_i|          j = argmax_{j in g} L_ij, u = y/|y|, v = P/|P|
//   dP_j  += k*gp_g(i) * (u_i - cos_ij v_j) / |P_j|                 accumulated per wave, block, then in order
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float* __restrict__ qry, int ldf,
                                                         const float* __restrict__ protos,
                                                         const float* __restrict__ dpred, float* __restrict__ dqry,
                                                         int ldd, float* __restrict__ part, int n, int c, int p,
                                                         float scalar, int* __restrict__ winners) {
    __shared__ float pn[MAXJ][64 * MAXCL];
    __shared__ float nrm[MAXJ];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int J = 2 * p;
    const float* pb = protos + (size_t)b * J * c;
    for (int j = wave; j < J; j += 4) {
        float s = 0.f;
        for (int ch = lane; ch < c; ch += 64) {
            float v = pb[(size_t)j * c + ch];
            s += v * v;
        }
        s = wave_sum(s);
        if (lane == 0) nrm[j] = fmaxf(sqrtf(s), 1e-8f);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < J * c; t += 256) {
        int j = t / c, ch = t - j * c;
        pn[j][ch] = pb[t] / nrm[j];
    }
    __syncthreads();
    const int ncl = (c + 255) / 256;
    float accP[MAXJ][MAXCL];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
#pragma unroll
        for (int e = 0; e < MAXCL; ++e) accP[j][e] = 0.f;

    // the next pixel's row is requested before the current one is worked on (the loop is latency-bound)
    float4 nxt[MAXCL / 4];
    auto fetch = [&](int i) {
        const float* xp = qry + ((size_t)b * n + min(i, n - 1)) * ldf;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            nxt[t] = (t < ncl && ch < c) ? *(const float4*)(xp + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(blockIdx.x * 4 + wave);
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        float u[MAXCL];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            const float4 v = nxt[t];
            u[t * 4 + 0] = v.x; u[t * 4 + 1] = v.y; u[t * 4 + 2] = v.z; u[t * 4 + 3] = v.w;
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        fetch(i + gridDim.x * 4);
        ss = wave_sum(ss);
        const float nx = fmaxf(sqrtf(ss), 1e-8f), rnx = 1.f / nx;
        float dot[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) dot[j] = 0.f;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u[t * 4 + e] = u[t * 4 + e] * rnx;
                if (t < ncl && ch < c) {
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j)
                        if (j < J) dot[j] += u[t * 4 + e] * pn[j][ch + e];
                }
            }
        }
        {
            const float tot = wave_sum8(dot);
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) dot[j] = __shfl(tot, j, 64);      // unscaled cosines
        }
        // winners (first maximum), fg rows [0,p) -> pred channel 1, bg rows [p,2p) -> channel 0
        int sel[2];
        float coef[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            int bj = g * p;
            float best = dot[g * p];
            for (int j = 1; j < p; ++j)
                if (dot[g * p + j] > best) {
                    best = dot[g * p + j];
                    bj = g * p + j;
                }
            sel[g] = bj;
            coef[g] = scalar * dpred[((size_t)b * 2 + (g == 0 ? 1 : 0)) * n + i];
            if (lane == 0) winners[((size_t)b * 2 + g) * n + i] = bj;
        }
        float dy[MAXCL];
#pragma unroll
        for (int e = 0; e < MAXCL; ++e) dy[e] = 0.f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                if (j < J && j == sel[g]) {      // wave-uniform
                    const float cj = dot[j], kk = coef[g], inp = 1.f / nrm[j];
#pragma unroll
                    for (int t = 0; t < MAXCL / 4; ++t) {
                        int ch = t * 256 + lane * 4;
                        if (t < ncl && ch < c) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = pn[j][ch + e], uu = u[t * 4 + e];
                                dy[t * 4 + e] += kk * (v - cj * uu) * rnx;
                                accP[j][t * 4 + e] += kk * (uu - cj * v) * inp;
                            }
                        }
                    }
                }
            }
        }
        float* dq = dqry + ((size_t)b * n + i) * ldd;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            if (t < ncl && ch < c) *(float4*)(dq + ch) = make_float4(dy[t * 4], dy[t * 4 + 1], dy[t * 4 + 2], dy[t * 4 + 3]);
        }
    }
    // block partial of dP: the four waves add in order through LDS (pn is free now)
    __syncthreads();
    float* red = &pn[0][0];      // [J][c] as 4 rounds
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int j = 0; j < MAXJ; ++j)
                if (j < J) {
#pragma unroll
                    for (int t = 0; t < MAXCL / 4; ++t) {
                        int ch = t * 256 + lane * 4;
                        if (t < ncl && ch < c) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float prev = wv == 0 ? 0.f : red[j * c + ch + e];
                                red[j * c + ch + e] = prev + accP[j][t * 4 + e];
                            }
                        }
                    }
                }
        }
        __syncthreads();
    }
    float* out = part + ((size_t)b * gridDim.x + blockIdx.x) * J * c;
    for (int t = threadIdx.x; t < J * c; t += 256) out[t] = red[t];
}

// out[k] = sum_{q<nparts} part[q][k]   (fixed order: wave v adds parts v, v+4, ..., then (s0+s1)+(s2+s3));
// optional transposed store [len/J][J] <- [J][len/J].  Block = 64 values x 4 part lanes.
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, int nparts, int len,
                                                        float* __restrict__ out, int transpose_J) {
    __shared__ float red[4][64];
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), kl = min(k, len - 1);
    const int grp = blockIdx.y;
    const float s = chunk_sum(part + (size_t)grp * nparts * len + kl, (size_t)len, nparts, red);
    if (threadIdx.x >= 64 || k >= len) return;
    if (transpose_J > 0) {
        const int cdim = len / transpose_J;
        const int j = k / cdim, ch = k - j * cdim;
        out[(size_t)grp * len + ch * transpose_J + j] = s;
    } else {
        out[(size_t)grp * len + k] = s;
    }
}

// -----------------------------------------------------------------------------------------------
// meta-prototype backward.  One wave per support pixel, block per (pixel group, bs).
//   dN_j = dP_j / (S D_sj),  dD_j = -(dP_j . P_sj) / (S D_sj),  da_ij = x_i . dN_j + dD_j
//   dsig_ij = da_ij m_g(i),  dd_ij = sig_ij (dsig_ij - sum_{k in g} sig_ik dsig_ik)
//   dx_i = sum_j a_ij dN_j - 2 sum_j dd_ij (x_i - ctr_j),   dctr_j += 2 sum_i dd_ij (x_i - ctr_j)
// MAP mode (p == 0 on entry -> J = 2): a = mask at feature resolution, or, for the Baseline's pooling over
// up-sampled features, the adjoint weights Aext[bs][g][i] the forward left in its workspace:
// dx_i = sum_g a_g(i) dN_g.
template <bool MPM>
__global__ __launch_bounds__(256) void mpm_bwd_kernel(const float* __restrict__ feat, int ldf,
                                                      const float* __restrict__ mask, const float* __restrict__ Aext,
                                                      const float* __restrict__ ctr,
                                                      const float* __restrict__ dP, const float* __restrict__ Pps,
                                                      const float* __restrict__ Dps, float* __restrict__ dsup, int ldd,
                                                      float* __restrict__ part, int S, int n, int h, int w, int H, int W,
                                                      int c, int p) {
    __shared__ float dN[MAXJ][64 * MAXCL];
    __shared__ float dD[MAXJ];
    const int bs = blockIdx.y, b = bs / S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int J = MPM ? 2 * p : 2;
    const int ncl = (c + 255) / 256;
    for (int j = wave; j < J; j += 4) {
        const float* dp = dP + ((size_t)b * J + j) * c;
        const float* pp = Pps + ((size_t)bs * J + j) * c;
        float s = 0.f;
        for (int ch = lane; ch < c; ch += 64) s += dp[ch] * pp[ch];
        s = wave_sum(s);
        if (lane == 0) dD[j] = -s / ((float)S * Dps[bs * J + j]);
    }
    for (int t = threadIdx.x; t < J * c; t += 256) {
        int j = t / c, ch = t - j * c;
        dN[j][ch] = dP[((size_t)b * J + j) * c + ch] / ((float)S * Dps[bs * J + j]);
    }
    __syncthreads();

    float cw[MAXJ][MAXCL];
    float accC[MAXJ][MAXCL];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
#pragma unroll
        for (int e = 0; e < MAXCL; ++e) {
            accC[j][e] = 0.f;
            cw[j][e] = 0.f;
        }
    if (MPM) {
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int ch = t * 256 + lane * 4 + e;
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) cw[j][t * 4 + e] = (t < ncl && ch < c && j < J) ? ctr[ch * J + j] : 0.f;
            }
    }
    const float* mk = mask + (size_t)bs * 2 * H * W;
    float4 nxt[MAXCL / 4];               // next pixel's row, requested one iteration ahead
    auto fetch = [&](int i) {
        const float* xp = feat + ((size_t)bs * n + min(i, n - 1)) * ldf;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            nxt[t] = (t < ncl && ch < c) ? *(const float4*)(xp + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(blockIdx.x * 4 + wave);
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        const int y = i / w, x = i - y * w;
        const int sy = nearest_src(y, H, h), sx = nearest_src(x, W, w);
        const float mg[2] = {mk[(size_t)sy * W + sx], mk[(size_t)H * W + (size_t)sy * W + sx]};
        float xv[MAXCL];
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            const float4 v = nxt[t];
            xv[t * 4 + 0] = v.x; xv[t * 4 + 1] = v.y; xv[t * 4 + 2] = v.z; xv[t * 4 + 3] = v.w;
        }
        fetch(i + gridDim.x * 4);
        float a[MAXJ], dd[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            a[j] = 0.f;
            dd[j] = 0.f;
        }
        // t_j = x . dN_j
        float tj[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) tj[j] = 0.f;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            if (t < ncl && ch < c) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j)
                        if (j < J) tj[j] += xv[t * 4 + e] * dN[j][ch + e];
            }
        }
        {
            const float tot = wave_sum8(tj);
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) tj[j] = __shfl(tot, j, 64);
        }
        if (MPM) {
            float d[MAXJ];
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) d[j] = 0.f;
#pragma unroll
            for (int t = 0; t < MAXCL / 4; ++t) {
                int ch = t * 256 + lane * 4;
                if (t < ncl && ch < c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < MAXJ; ++j) {
                            float df = xv[t * 4 + e] - cw[j][t * 4 + e];
                            d[j] += df * df;
                        }
                }
            }
            {
                const float tot = wave_sum8(d);
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) d[j] = -__shfl(tot, j, 64);
            }
            for (int g = 0; g < 2; ++g) {
                float mx = -INFINITY;
                for (int j = 0; j < p; ++j) mx = fmaxf(mx, d[g * p + j]);
                float sg[MAXJ / 2], ssum = 0.f;
                for (int j = 0; j < p; ++j) {
                    sg[j] = expf(d[g * p + j] - mx);
                    ssum += sg[j];
                }
                float dot = 0.f, ds[MAXJ / 2];
                for (int j = 0; j < p; ++j) {
                    sg[j] /= ssum;
                    ds[j] = (tj[g * p + j] + dD[g * p + j]) * mg[g];
                    dot += sg[j] * ds[j];
                }
                for (int j = 0; j < p; ++j) {
                    a[g * p + j] = sg[j] * mg[g];
                    dd[g * p + j] = sg[j] * (ds[j] - dot);
                }
            }
        } else if (Aext) {
            a[0] = Aext[((size_t)bs * 2 + 0) * n + i];
            a[1] = Aext[((size_t)bs * 2 + 1) * n + i];
        } else {
            a[0] = mg[0];
            a[1] = mg[1];
        }
        float* dx = dsup + ((size_t)bs * n + i) * ldd;
#pragma unroll
        for (int t = 0; t < MAXCL / 4; ++t) {
            int ch = t * 256 + lane * 4;
            if (t < ncl && ch < c) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j)
                        if (j < J) {
                            s += a[j] * dN[j][ch + e];
                            if (MPM) {
                                const float df = xv[t * 4 + e] - cw[j][t * 4 + e];
                                s -= 2.f * dd[j] * df;
                                accC[j][t * 4 + e] += 2.f * dd[j] * df;
                            }
                        }
                    o[e] = s;
                }
                *(float4*)(dx + ch) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    if (!MPM) return;
    __syncthreads();
    float* red = &dN[0][0];
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int j = 0; j < MAXJ; ++j)
                if (j < J) {
#pragma unroll
                    for (int t = 0; t < MAXCL / 4; ++t) {
                        int ch = t * 256 + lane * 4;
                        if (t < ncl && ch < c) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float prev = wv == 0 ? 0.f : red[j * c + ch + e];
                                red[j * c + ch + e] = prev + accC[j][t * 4 + e];
                            }
                        }
                    }
                }
        }
        __syncthreads();
    }
    float* out = part + ((size_t)bs * gridDim.x + blockIdx.x) * J * c;
    for (int t = threadIdx.x; t < J * c; t += 256) out[t] = red[t];
}

constexpr int HB_BLOCKS = 128;   // pixel-group blocks per image in the two wave-per-pixel kernels

}  // namespace pemp

using namespace pemp;

// workspace (fp32): Pps[BS][J][c] | Dps[BS][J] | dpred[B][2][n] | dP[B][J][c] | cpart[B][HB][J][c] | mpart[BS][HB][J][c] |
// winners int32 [B][2][n] (the LAST B * 2 * n words: the prototype row the cosine backward routed each (query pixel, group)
// gradient to -- group 0 = foreground rows [0, p), group 1 = background rows [p, 2p); what a decision-frozen reference
// evaluation needs to know, tests/test_grad_frozen_gpu.py)
static size_t head_bwd_floats(int B, int S, int n, int c, int J) {
    const size_t BS = (size_t)B * S;
    return BS * J * c + BS * J + (size_t)B * 2 * n + (size_t)B * J * c + (size_t)B * HB_BLOCKS * J * c +
           BS * HB_BLOCKS * J * c + 64 + (size_t)B * 2 * n;
}

extern "C" size_t pemp_head_bwd_workspace_bytes(int B, int S, int n, int c, int p) {
    return head_bwd_floats(B, S, n, c, p > 0 ? 2 * p : 2) * sizeof(float);
}

// Gradient of mean cross-entropy w.r.t. the support/query features and ctr.
//   fwd_ws   the workspace pemp_mpm_protos_f32 / pemp_masked_avg_pool_f32(full_res=0) left behind for the
//            same inputs (holds the pooling partial sums)
//   protos   [B][J][c] prototypes of the forward; pred [B][2][n] its low-res prediction
//   target   int64 [B][Ho][Wo]; weight [B][Ho][Wo] per-pixel CE weights or NULL (CELossDT);
//   stats    [B][8] from pemp_eval_tail(_weighted)_f32 (loss denominator per episode at index 1)
//   dsup [B*S][n][ldd], dqry [B][n][ldd] out; dctr [c][2p] out (ignored when p == 0: plain MAP)
static int head_bwd_impl(const float* sup_feat, const float* qry_feat, int ldf, const float* mask,
                         const float* ctr, const void* fwd_ws, const float* protos, const float* pred,
                         const int64_t* target, const float* weight, const double* stats, const float* dlogits,
                         float* dsup, float* dqry, int ldd,
                         float* dctr, void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W,
                         int Ho, int Wo, int c, int p, int map_full_res, float dist_scalar, void* stream) {
    PEMP_REQUIRE(sup_feat && qry_feat && mask && fwd_ws && protos && dsup && dqry && ws &&
                     (dlogits || (pred && target && stats)),
                 "head_bwd: null pointer");
    PEMP_REQUIRE(B > 0 && S > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0 && p >= 0 && 2 * p <= MAXJ, "head_bwd: bad dims");
    PEMP_REQUIRE(c > 0 && c % 4 == 0 && c <= 64 * MAXCL && ldf >= c && ldd >= c && ldf % 4 == 0 && ldd % 4 == 0,
                 "head_bwd: c=%d must be a multiple of 4 and <= %d", c, 64 * MAXCL);
    PEMP_REQUIRE(p == 0 || (ctr && dctr), "head_bwd: ctr/dctr required when p > 0");
    PEMP_REQUIRE(!map_full_res || p == 0, "head_bwd: map_full_res is the Baseline's plain-MAP head (p == 0)");
    const int n = h * w, BS = B * S, J = p > 0 ? 2 * p : 2;
    PEMP_REQUIRE(ws_bytes >= head_bwd_floats(B, S, n, c, J) * sizeof(float), "head_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const PoolWs L = pool_ws_layout(const_cast<void*>(fwd_ws), BS, n, c, J);
    float* Pps = (float*)ws;
    float* Dps = Pps + (size_t)BS * J * c;
    float* dpred = Dps + (size_t)BS * J;
    float* dP = dpred + (size_t)B * 2 * n;
    float* cpart = dP + (size_t)B * J * c;
    float* mpart = cpart + (size_t)B * HB_BLOCKS * J * c;
    const int nck = nchunks_of(n);
    hipLaunchKernelGGL(pool_shot_kernel, dim3(J, BS, cdiv(c, 64)), dim3(256), 0, st, (const float*)L.part,
                       (const float*)L.asum, map_full_res ? (const float*)L.msum : (const float*)nullptr, Pps, Dps, c, J,
                       nck, p > 0 ? 1e-6f : 1e-5f);
    if (dlogits)
        hipLaunchKernelGGL(upsample_bwd_kernel<false>, dim3(cdiv(n, 4), B), dim3(256), 0, st, (const float*)nullptr,
                           (const int64_t*)nullptr, (const float*)nullptr, (const double*)nullptr, B, dlogits, dpred, h, w, Ho,
                           Wo);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel<true>, dim3(cdiv(n, 4), B), dim3(256), 0, st, pred, target, weight, stats, B,
                           (const float*)nullptr, dpred, h, w, Ho, Wo);
    hipLaunchKernelGGL(cosine_bwd_kernel, dim3(HB_BLOCKS, B), dim3(256), 0, st, qry_feat, ldf, protos, (const float*)dpred,
                       dqry, ldd, cpart, n, c, p > 0 ? p : 1, dist_scalar,
                       (int*)((float*)ws + head_bwd_floats(B, S, n, c, J) - (size_t)B * 2 * n));
    hipLaunchKernelGGL(sum_parts_kernel, dim3(cdiv(J * c, 64), B), dim3(256), 0, st, (const float*)cpart, HB_BLOCKS, J * c,
                       dP, 0);
    int e = launch_status("head_bwd/cosine");
    if (e) return e;
    if (p > 0) {
        hipLaunchKernelGGL(mpm_bwd_kernel<true>, dim3(HB_BLOCKS, BS), dim3(256), 0, st, sup_feat, ldf, mask,
                           (const float*)nullptr, ctr,
                           (const float*)dP, (const float*)Pps, (const float*)Dps, dsup, ldd, mpart, S, n, h, w, H, W, c, p);
        hipLaunchKernelGGL(sum_parts_kernel, dim3(cdiv(J * c, 64), 1), dim3(256), 0, st, (const float*)mpart,
                           BS * HB_BLOCKS, J * c, dctr, J);
    } else {
        hipLaunchKernelGGL(mpm_bwd_kernel<false>, dim3(HB_BLOCKS, BS), dim3(256), 0, st, sup_feat, ldf, mask,
                           map_full_res ? (const float*)L.A : (const float*)nullptr, ctr,
                           (const float*)dP, (const float*)Pps, (const float*)Dps, dsup, ldd, mpart, S, n, h, w, H, W, c, 1);
    }
    return launch_status("head_bwd/mpm");
}

extern "C" int pemp_head_bwd_f32(const float* sup_feat, const float* qry_feat, int ldf, const float* mask,
                                 const float* ctr, const void* fwd_ws, const float* protos, const float* pred,
                                 const int64_t* target, const float* weight, const double* stats, float* dsup,
                                 float* dqry, int ldd,
                                 float* dctr, void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W,
                                 int Ho, int Wo, int c, int p, int map_full_res, float dist_scalar, void* stream) {
    PEMP_REQUIRE(pred && target && stats, "head_bwd: null pointer");
    return head_bwd_impl(sup_feat, qry_feat, ldf, mask, ctr, fwd_ws, protos, pred, target, weight, stats, nullptr, dsup, dqry,
                         ldd, dctr, ws, ws_bytes, B, S, h, w, H, W, Ho, Wo, c, p, map_full_res, dist_scalar, stream);
}

// The same backward for an ARBITRARY gradient of the logits (autograd hands dL/dlogits to the model's output):
// dlogits [B][2][Ho][Wo] replaces (pred, target, weight, stats).
extern "C" int pemp_head_bwd_dlogits_f32(const float* sup_feat, const float* qry_feat, int ldf, const float* mask,
                                         const float* ctr, const void* fwd_ws, const float* protos, const float* dlogits,
                                         float* dsup, float* dqry, int ldd, float* dctr, void* ws, size_t ws_bytes, int B,
                                         int S, int h, int w, int H, int W, int Ho, int Wo, int c, int p, int map_full_res,
                                         float dist_scalar, void* stream) {
    PEMP_REQUIRE(dlogits, "head_bwd_dlogits: null pointer");
    return head_bwd_impl(sup_feat, qry_feat, ldf, mask, ctr, fwd_ws, protos, nullptr, nullptr, nullptr, nullptr, dlogits, dsup,
                         dqry, ldd, dctr, ws, ws_bytes, B, S, h, w, H, W, Ho, Wo, c, p, map_full_res, dist_scalar, stream);
}
