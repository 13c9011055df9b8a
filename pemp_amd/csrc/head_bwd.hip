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


// Compiled twice, as head.hip: MAXJ = 8 (protos <= 4) and MAXJ = 16 (protos 5..8); the entry points pick by 2p.
#define PEMP_MAXJ 8
#define PEMP_HEAD_NS j8
#include "head_bwd_body.inc"
#undef PEMP_MAXJ
#undef PEMP_HEAD_NS
#define PEMP_MAXJ 16
#define PEMP_HEAD_NS j16
#include "head_bwd_body.inc"
#undef PEMP_MAXJ
#undef PEMP_HEAD_NS

using namespace pemp;

extern "C" size_t pemp_head_bwd_workspace_bytes(int B, int S, int n, int c, int p) { return j8::pemp_head_bwd_workspace_bytes(B, S, n, c, p); }

extern "C" int pemp_head_bwd_f32(const float* sup_feat, const float* qry_feat, int ldf, const float* mask,
                                 const float* ctr, const void* fwd_ws, const float* protos, const float* pred,
                                 const int64_t* target, const float* weight, const double* stats, float* dsup,
                                 float* dqry, int ldd,
                                 float* dctr, void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W,
                                 int Ho, int Wo, int c, int p, int map_full_res, float dist_scalar, void* stream) {
    if (2 * p <= 8)
        return j8::pemp_head_bwd_f32(sup_feat, qry_feat, ldf, mask, ctr, fwd_ws, protos, pred, target, weight, stats, dsup, dqry, ldd,
                                     dctr, ws, ws_bytes, B, S, h, w, H, W, Ho, Wo, c, p, map_full_res, dist_scalar, stream);
    return j16::pemp_head_bwd_f32(sup_feat, qry_feat, ldf, mask, ctr, fwd_ws, protos, pred, target, weight, stats, dsup, dqry, ldd,
                                  dctr, ws, ws_bytes, B, S, h, w, H, W, Ho, Wo, c, p, map_full_res, dist_scalar, stream);
}

extern "C" int pemp_head_bwd_dlogits_f32(const float* sup_feat, const float* qry_feat, int ldf, const float* mask,
                                         const float* ctr, const void* fwd_ws, const float* protos, const float* dlogits,
                                         float* dsup, float* dqry, int ldd, float* dctr, void* ws, size_t ws_bytes, int B,
                                         int S, int h, int w, int H, int W, int Ho, int Wo, int c, int p, int map_full_res,
                                         float dist_scalar, void* stream) {
    if (2 * p <= 8)
        return j8::pemp_head_bwd_dlogits_f32(sup_feat, qry_feat, ldf, mask, ctr, fwd_ws, protos, dlogits, dsup, dqry, ldd, dctr, ws,
                                             ws_bytes, B, S, h, w, H, W, Ho, Wo, c, p, map_full_res, dist_scalar, stream);
    return j16::pemp_head_bwd_dlogits_f32(sup_feat, qry_feat, ldf, mask, ctr, fwd_ws, protos, dlogits, dsup, dqry, ldd, dctr, ws,
                                          ws_bytes, B, S, h, w, H, W, Ho, Wo, c, p, map_full_res, dist_scalar, stream);
}
