// Prototype head of PEMP on gfx950: meta-prototype module (soft assignment + masked pooling),
// plain / full-resolution masked average pooling, pixel x prototype cosine map with group max,
// and the upsample / argmax / cross-entropy / IoU-count tail of the evaluator.
//
// All of these are streaming kernels over [pixels][c] feature maps (c contiguous): their bound is HBM
// bandwidth (AI ~ 3 flop/B at 2p = 6 prototypes).  The three passes over the features run on the matrix
// cores -- per-pixel products with a c x 2p table as 16x16x4 fp32 MFMA row streams (cosine map, MPM
// assignment), masked pooling as an MFMA over pixels -- with lane-contiguous loads, 8-16 KB per wave in
// flight and at most 96 VGPRs / 32 KB of LDS per block so that five blocks share a CU (measured 5.4-5.7
// TB/s).  Wave-per-pixel VALU variants remain for channel counts the MFMA kernels are not instantiated
// for.  Partial sums are combined in a fixed order everywhere (deterministic).
#include <stdlib.h>
#include "head_common.h"

// The kernels are written with a compile-time bound MAXJ on the rows per pixel (2 * protos) they keep in registers.  The body is
// compiled twice: MAXJ = 8 (protos <= 4: the reference's default 3; MFMA row-stream kernels) and MAXJ = 16 (protos 5..8: the
// wave-per-pixel kernels only); the entry points pick by 2p.
#define PEMP_MAXJ 8
#define PEMP_HEAD_NS j8
#include "head_body.inc"
#undef PEMP_MAXJ
#undef PEMP_HEAD_NS
#define PEMP_MAXJ 16
#define PEMP_HEAD_NS j16
#include "head_body.inc"
#undef PEMP_MAXJ
#undef PEMP_HEAD_NS

using namespace pemp;

extern "C" size_t pemp_mpm_workspace_bytes(int B, int S, int n, int c, int p) { return j8::pemp_mpm_workspace_bytes(B, S, n, c, p); }
extern "C" size_t pemp_map_workspace_bytes(int B, int S, int n, int c) { return j8::pemp_map_workspace_bytes(B, S, n, c); }

extern "C" int pemp_mpm_protos_f32(const float* feat, int ldf, const float* mask, const float* ctr, float* protos,
                                   void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W, int c, int p,
                                   void* stream) {
    if (2 * p <= 8) return j8::pemp_mpm_protos_f32(feat, ldf, mask, ctr, protos, ws, ws_bytes, B, S, h, w, H, W, c, p, stream);
    return j16::pemp_mpm_protos_f32(feat, ldf, mask, ctr, protos, ws, ws_bytes, B, S, h, w, H, W, c, p, stream);
}

extern "C" int pemp_masked_avg_pool_f32(const float* feat, int ldf, const float* mask, float* protos, void* ws,
                                        size_t ws_bytes, int B, int S, int h, int w, int H, int W, int c, int full_res,
                                        void* stream) {
    return j8::pemp_masked_avg_pool_f32(feat, ldf, mask, protos, ws, ws_bytes, B, S, h, w, H, W, c, full_res, stream);
}

extern "C" int pemp_cosine_proto_max_f32(const float* qry, int ldf, const float* protos, float* pred, uint8_t* resp,
                                         int B, int n, int c, int p, float dist_scalar, void* stream) {
    if (2 * p <= 8) return j8::pemp_cosine_proto_max_f32(qry, ldf, protos, pred, resp, B, n, c, p, dist_scalar, stream);
    return j16::pemp_cosine_proto_max_f32(qry, ldf, protos, pred, resp, B, n, c, p, dist_scalar, stream);
}

extern "C" int pemp_argmax_masks_f32(const float* pred, float* masks, int B, int n, void* stream) {
    return j8::pemp_argmax_masks_f32(pred, masks, B, n, stream);
}

extern "C" int pemp_upsample_bilinear_ac_f32(const float* pred, float* out, int B, int C, int h, int w, int Ho, int Wo,
                                             void* stream) {
    return j8::pemp_upsample_bilinear_ac_f32(pred, out, B, C, h, w, Ho, Wo, stream);
}

extern "C" int pemp_upsample_nearest_u8_i64(const uint8_t* resp, int64_t* out, int B, int h, int w, int Ho, int Wo,
                                            void* stream) {
    return j8::pemp_upsample_nearest_u8_i64(resp, out, B, h, w, Ho, Wo, stream);
}

extern "C" size_t pemp_eval_tail_workspace_bytes(int B, int Ho, int Wo) { return j8::pemp_eval_tail_workspace_bytes(B, Ho, Wo); }

extern "C" int pemp_eval_tail_weighted_f32(const float* pred, const int64_t* target, const float* weight,
                                           uint8_t* pred_out, float* logits_out, double* stats, void* ws,
                                           size_t ws_bytes, int B, int h, int w, int Ho, int Wo, void* stream) {
    return j8::pemp_eval_tail_weighted_f32(pred, target, weight, pred_out, logits_out, stats, ws, ws_bytes, B, h, w, Ho, Wo, stream);
}

extern "C" int pemp_eval_tail_f32(const float* pred, const int64_t* target, uint8_t* pred_out, float* logits_out,
                                  double* stats, void* ws, size_t ws_bytes, int B, int h, int w, int Ho, int Wo,
                                  void* stream) {
    return j8::pemp_eval_tail_weighted_f32(pred, target, nullptr, pred_out, logits_out, stats, ws, ws_bytes, B, h, w, Ho, Wo, stream);
}
