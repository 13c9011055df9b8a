/*
 * pemp_hip.h -- C ABI of libpemp_hip.so: the MI355X (gfx950) kernels of the PEMP
 * prototype-matching hot path.
 *
 * The reference (Jarvis73/PEMP) has no native/FFI interface: its hot path is a chain of stock
 * ATen ops issued from networks/{backbones,pemp_stage1,pemp_stage2,baseline}.py.  Each entry
 * point below names the reference call site(s) (file:line, relative to the reference root) whose
 * arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless the name/type says otherwise;
 *  - activations are NHWC ("pixel-major": [N][H][W][C], C contiguous) with an explicit per-pixel
 *    stride `ld*` in elements, so a kernel can read/write a channel slice of a wider buffer;
 *  - conv weights are "KRSC": [Cout][KH][KW][Cin], Cin contiguous (the GEMM K axis);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *  - no hidden allocation, no global mutable state; re-entrant across streams and devices;
 *  - return value: 0 ok, <0 invalid argument (see pemp_last_error()), >0 a hipError_t.
 */
#ifndef PEMP_HIP_H
#define PEMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEMP_ABI_VERSION 2   /* 2: pemp_adam_clip_step_f32 takes its hyper-parameters as doubles (round 5) */

/* flags for pemp_conv_desc.flags */
#define PEMP_CONV_RELU 1u          /* y = max(y, 0) after affine (+ residual)            */
#define PEMP_CONV_SHIFT_PER_IMAGE 2u /* shift is [N][Cout] instead of [Cout]              */
#define PEMP_CONV_STEM4 4u         /* input is NHWC4, K axis = taps x 4 channels (7x7 stem) */

typedef struct pemp_conv_desc {
    int32_t N, H, W;        /* input images, input spatial size                                  */
    int32_t Cin, ldx;       /* input channels read, input per-pixel stride (>= Cin)              */
    int32_t Ho, Wo;         /* output spatial size (caller computes; checked)                    */
    int32_t Cout, ldy;      /* output channels, output per-pixel stride (>= Cout)                */
    int32_t KH, KW;         /* kernel size                                                       */
    int32_t stride, pad, dil;
    int32_t ldr;            /* residual per-pixel stride (ignored when residual == NULL)         */
    int32_t Kpad;           /* weight row length in floats (>= KH*KW*Cin, multiple of 32)        */
    uint32_t flags;
    int32_t tile;           /* 0 = auto; 1 = 128x128, 2 = 128x64, 3 = 64x64 block tile (register staging);
                               11..13 the same with LDS-DMA staging; 14/15 = 128x128 / 128x64, 8 waves;
                               16/17 = 256x128 / 256x256, 8 waves; 21..27 = the shapes of 11..17 on the
                               buffer-addressed kernels; 28 = 32x64 blocks of 16-row wave tiles on
                               v_mfma_f32_16x16x4_f32 (few-row launches: finer granularity); 29 = hybrid of 23
                               and 28 in one grid for launches of a few rounds (other geometries: 23)
                               -- all bit-identical results; 31..37: split-K forms of 21..27 (see below) */
} pemp_conv_desc;

const char* pemp_last_error(void);
int pemp_abi_version(void);

/* Convolution + per-channel affine (+ residual) (+ ReLU) as one implicit GEMM on
 * v_mfma_f32_32x32x2_f32:   y[n,ho,wo,co] = act( scale[co] * sum_{kh,kw,ci} x[...] * w[co,kh,kw,ci]
 *                                               + shift[co] (+ residual[n,ho,wo,co]) )
 * scale == NULL means 1, shift == NULL means 0.
 * Replaces nn.Conv2d + nn.BatchNorm2d(eval) + ReLU + residual add:
 *   networks/backbones.py:47-52,66-75 (BottleNeck), :89-91 (stem), :110-111 (downsample),
 *   :330-357,366 (ASPPV2 convs + layer6), :281-305,319 (ASPP), :375-397 (VGG16);
 *   networks/pemp_stage1.py:74,77 (purifier); networks/baseline.py:61 (projection).        */
int pemp_conv2d_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y,
                         const float* scale, const float* shift, const float* residual,
                         void* stream);

/* Same convolution with a per-input-channel PADDING VALUE: out-of-image taps read pad_value[ci] instead
 * of 0 (pad_value NULL = the call above).  This is what lets a BatchNorm that sits IN FRONT of a padded
 * conv (ASPPV2 branches, networks/backbones.py:330-357: BN -> ReLU-less conv on the BN output, zero-padded
 * in BN space) fold into the conv exactly:  conv_W(s*x + t, pad 0) = conv_{W*s}(x, pad -t/s) + sum_taps W t.
 * Multi-tap, non-stem convs only.                                                                   */
int pemp_conv2d_padv_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y,
                              const float* scale, const float* shift, const float* residual,
                              const float* pad_value, void* stream);

/* Tile id 29 ("hybrid"): for a conv whose 32 x 32 wave tiles fill only a few rounds of the chip (a one-episode evaluation step's
 * 256-channel convs: 1304 tiles on 1024 SIMDs -- reference networks/backbones.py:42-77 at the reference's own data.test_bs = 1,
 * data_kits/datasets.py:23), the rows that fill WHOLE rounds run on the 64 x 64 tile and the remaining rows on 16-row wave tiles
 * in the same grid; bit-identical to every other exact variant.  Returns how many output rows of `d` go to the 64 x 64 part
 * (0: this geometry has no such split -- pemp_conv2d_nhwc_f32 then runs tile 29 as tile 23).                                */
int pemp_conv2d_hybrid_rows(const pemp_conv_desc* d);

/* Up to 4 INDEPENDENT convolutions in ONE launch (arrays of n descriptors / operand pointers; scale, shift, residual,
 * pad_value: NULL or arrays with NULL entries; pad_value for every member or for none; every descriptor names the same
 * tile variant 21..28).  Each member is computed exactly as by pemp_conv2d_padv_nhwc_f32 on its own -- same tiles, same K
 * order, bit-identical -- but the members' tiles share one grid: a one-episode evaluation step (5202 feature rows: 41-82
 * tiles per conv on 256 CUs) runs the dilated ASPPV2 branches (networks/backbones.py:330-357, all reading the same
 * activations) and a stage's downsample conv beside its conv1 (backbones.py:47,110) this way.  Members must not write
 * memory another member reads or writes.                                                                            */
int pemp_conv2d_group_nhwc_f32(int n, const pemp_conv_desc* d, const float* const* x, const float* const* w,
                               float* const* y, const float* const* scale, const float* const* shift,
                               const float* const* residual, const float* const* pad_value, void* stream);

/* SIDE-FIGURE VARIANT, not the product path's arithmetic: the same convolution with bf16 OPERANDS (x, w, residual, pad_value:
 * bf16 tensors; descriptor in bf16 elements, Cin % 64 == 0, ldx % 8 == 0) and fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16; y is bf16 (out_f32 == 0; a residual is then bf16 too) or fp32 (the encoder's last layer).  Tiles
 * 21..27 (0 = 24).  Exists to show what exact fp32 costs (bench.py: `bf16_variant`, with the mIoU it moves); never the default. */
int pemp_conv2d_bf16_nhwc(const pemp_conv_desc* d, const void* x, const void* w, void* y, const float* scale,
                          const float* shift, const void* residual, const void* pad_value, int out_f32, void* stream);
/* element-wise fp32 <-> bf16 (round to nearest even) for the boundaries of that variant (n % 4 == 0) */
int pemp_convert_f32_bf16(const float* x, void* y, long long n, void* stream);
int pemp_convert_bf16_f32(const void* x, float* y, long long n, void* stream);

/* The convolution above FOLLOWED BY DropBlock2D's scaling of its output (training; dropblock==0.3.0's two statements
 * y * mask[None] ; y * numel / sum(mask), in that order and rounding): row m of the result is multiplied by rowmask[m] (fp32 {0,1}
 * per output pixel, from pemp_dropblock_mask_f32) * M / *kept_count.  One launch instead of the conv + pemp_pixel_scale_f32 --
 * the purifier's conv -> ReLU -> DropBlock (networks/pemp_stage1.py:74-79) and, in the backward pass, the input-gradient convs
 * whose result DropBlock's backward scales the same way.  Tiles 21..27, or 31..37 with a split-K workspace (ws may be NULL
 * otherwise).                                                                                                              */
int pemp_conv2d_dropblock_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* scale,
                                   const float* shift, const float* residual, const float* rowmask, const int* kept_count,
                                   void* ws, size_t ws_bytes, void* stream);
/* train-mode BatchNorm apply (batch statistics given) + the DropBlock2D behind it in one pass: ASPPV2's BatchNorm -> DropBlock ->
 * conv branches (networks/backbones.py:329-353).                                                                           */
int pemp_bn_apply_dropblock_f32(const float* z, int ldz, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, float* y, int ldy, int M, int C, const float* rowmask,
                                const int* kept_count, void* stream);

/* [N,3,H,W] image (+ optional [N,1,H,W] prior; NULL -> 0) -> NHWC4 [N,H,W,4].
 * Replaces torch.cat/view at networks/pemp_stage1.py:139, pemp_stage2.py:130-138.          */
int pemp_pack_input_nhwc4_f32(const float* img, const float* prior, float* out,
                              int N, int H, int W, void* stream);

/* nn.MaxPool2d(k, s, p, ceil_mode) on NHWC: networks/backbones.py:92 (3/2/1 ceil), :378-392.
 * Ho/Wo are given by the caller (checked against the formula).                              */
int pemp_maxpool2d_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int ldx,
                            int Ho, int Wo, int ldy, int k, int s, int p, void* stream);

/* Training form of the same layer (contiguous x, y): also writes, per output element, the offset
 * dh*k+dw of the winning element inside its unclipped window (first maximum in scan order, as ATen's
 * max_pool2d_with_indices), one byte each; pemp_maxpool2d_idx_bwd_nhwc_f32 routes dy through those
 * indices (backward of backbones.py:92) without reading x again.                              */
int pemp_maxpool2d_idx_nhwc_f32(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C,
                                int Ho, int Wo, int k, int s, int p, void* stream);
int pemp_maxpool2d_idx_bwd_nhwc_f32(const uint8_t* idx, const float* dy, float* dx, int N, int H, int W,
                                    int C, int Ho, int Wo, int k, int s, int p, void* stream);

/* F.adaptive_avg_pool2d(x,(1,1)) on NHWC: y[n][c] = mean_i x[n][i][c]
 * (networks/backbones.py:311,360).                                                          */
int pemp_global_avgpool_nhwc_f32(const float* x, float* y, int N, int HW, int C, int ldx,
                                 void* stream);

/* y[m][c] = x[m][c]*scale[c] + shift[c] for `nb` (scale,shift,y) triples sharing one x
 * (the BatchNorm that PRECEDES each ASPPV2 branch conv: networks/backbones.py:328,334,340,346,352). */
int pemp_channel_affine_multi_f32(const float* x, int ldx, int M, int C, int nb,
                                  const float* const* scale, const float* const* shift,
                                  float* const* y, const int* ldy, void* stream);

/* Meta-prototype module, steps (i)-(v) of networks/pemp_stage1.py:201-213 (= pemp_stage2.py:170-186):
 * soft-assign every support pixel to the p fg / p bg centres, masked, and pool.
 *   feat   [B*S][n][ldf] support features (NHWC, c channels), n = h*w
 *   mask   [B*S][2][H][W] full-resolution support mask (ch0 fg, ch1 bg); resized on the fly with
 *          F.interpolate(..., mode="nearest") semantics (pemp_stage1.py:147)
 *   ctr    [c][2p] learnable centres (pemp_stage1.py:104-105)
 *   protos [B][2p][c] out: row j<p = fg prototype j, row p+j = bg prototype j (mean over S)
 *   ws     workspace of pemp_mpm_workspace_bytes() bytes                                     */
size_t pemp_mpm_workspace_bytes(int B, int S, int n, int c, int p);
int pemp_mpm_protos_f32(const float* feat, int ldf, const float* mask, const float* ctr,
                        float* protos, void* ws, size_t ws_bytes,
                        int B, int S, int h, int w, int H, int W, int c, int p, void* stream);

/* Plain masked average pooling (protos == 0 branch, networks/pemp_stage1.py:223-227) when
 * full_res == 0, and the Baseline form over bilinearly upsampled features
 * (networks/baseline.py:100-110) when full_res == 1 (computed through the adjoint of the
 * align_corners=True interpolation, never materialising the upsampled tensor).
 *   protos [B][2][c] out: row 0 = fg, row 1 = bg                                             */
size_t pemp_map_workspace_bytes(int B, int S, int n, int c);
int pemp_masked_avg_pool_f32(const float* feat, int ldf, const float* mask, float* protos,
                             void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W,
                             int c, int full_res, void* stream);

/* F.cosine_similarity(qry, proto) * dist_scalar for every prototype, max over the p prototypes
 * of each group, stacked (bg, fg): networks/pemp_stage1.py:214-215,256-260.
 *   qry    [B][n][ldf] query features (Q == 1, as the reference's broadcasting requires)
 *   protos [B][2p][c]  (layout of pemp_mpm_protos_f32 / pemp_masked_avg_pool_f32 with p = 1)
 *   pred   [B][2][n]   out, ch0 = bg, ch1 = fg
 *   resp   [B][n] uint8 out or NULL: response index of pemp_stage1.py:217-222 (0..2p-1)      */
int pemp_cosine_proto_max_f32(const float* qry, int ldf, const float* protos, float* pred,
                              uint8_t* resp, int B, int n, int c, int p, float dist_scalar,
                              void* stream);

/* PANet's alignment branch (networks/panet.py:168-171): masks [B][2][n] from a low-resolution prediction
 * pred [B][2][n]:  masks[b][0] = [argmax == 1] (foreground), masks[b][1] = [argmax == 0]; channel 0 wins ties
 * (torch.argmax).  They feed pemp_masked_avg_pool_f32(full_res = 0) with the QUERY features as "support".    */
int pemp_argmax_masks_f32(const float* pred, float* masks, int B, int n, void* stream);

/* F.interpolate(pred, (Ho,Wo), "bilinear", align_corners=True) -> logits [B][2][Ho][Wo]
 * (networks/pemp_stage1.py:157,162; baseline.py:117).                                        */
int pemp_upsample_bilinear_ac_f32(const float* pred, float* out, int B, int C, int h, int w,
                                  int Ho, int Wo, void* stream);

/* F.interpolate(resp.float(), (Ho,Wo), "nearest").long() (networks/pemp_stage1.py:158-159).   */
int pemp_upsample_nearest_u8_i64(const uint8_t* resp, int64_t* out, int B, int h, int w,
                                 int Ho, int Wo, void* stream);

/* The tail of Evaluator.test_step fused (entry/pemp_stage1.py:48-53 + core/metrics.py:9-23):
 * upsample (as above) + argmax over the 2 classes + CrossEntropyLoss(ignore_index=255) partial
 * sums + FewShotMetric tp/fp/fn for class rows {bg, fg}.
 *   target int64 [B][Ho][Wo]; pred_out uint8 [B][Ho][Wo];
 *   stats  double [B][8] out: {ce_sum, n_valid, tp_bg, fp_bg, fn_bg, tp_fg, fp_fg, fn_fg}
 *   logits_out [B][2][Ho][Wo] or NULL.                                                       */
size_t pemp_eval_tail_workspace_bytes(int B, int Ho, int Wo);
int pemp_eval_tail_f32(const float* pred, const int64_t* target, uint8_t* pred_out,
                       float* logits_out, double* stats, void* ws, size_t ws_bytes,
                       int B, int h, int w, int Ho, int Wo, void* stream);

/* ResNetCM.comm statistics (networks/backbones.py:208-216): mask' = max_pool2d(mask,3,stride,1);
 * mean over ALL pixels and max over pixels of x*mask' per image and channel.
 *   x [N][Hx*Wx][ldx], mask_in [N][Hm][Wm], mask_out [N][Hx][Wx], stat [N][2][C] (mean, max).
 * x == NULL pools the mask only (the extra pool at backbones.py:227).                          */
int pemp_cm_reduce_f32(const float* x, int ldx, const float* mask_in, float* mask_out, float* stat,
                       int N, int Hm, int Wm, int Hx, int Wx, int C, int stride, void* stream);

/* ResNetCM.comm after the statistics (networks/backbones.py:213-221): per episode g the mean over its S+Q images
 * of (mean, max), then Linear(2C -> 2):
 *   agg[g][k] = mean_{n: group[n]==g} stat[n][k];   feat[g][e] = lin_b[e] + sum_k agg[g][k] * lin_w[e][k]
 * stat [N][C2] (C2 = 2C: means then maxes, the layout pemp_cm_reduce_f32 writes), group int32 [N], G <= 64.      */
int pemp_cm_linear_f32(const float* stat, const int32_t* group, const float* lin_w, const float* lin_b,
                       float* agg, float* feat, int N, int G, int C2, void* stream);
/* The two broadcast channels enter the stage's first 1x1 convs (backbones.py:194,201,231,236,241) as a per-image bias:
 *   out[n][co] = base[co] + alpha[co] * (feat[g][0] * wext[co*ldw] + feat[g][1] * wext[co*ldw + 1]),  g = group[n]
 * (alpha/base: the folded eval BatchNorm, or NULL = 1 / 0 in training where BN follows separately).             */
int pemp_cm_bias_f32(const float* feat, const int32_t* group, const float* wext, int ldw, const float* alpha,
                     const float* base, float* out, int N, int Cout, void* stream);
/* Backward of the two (entry/pemp_stage2.py:78).  colsum [N][Cout] = per-image sums of the conv-output gradient:
 *   dwext[co][e] = sum_n colsum[n][co] * feat[g(n)][e];   dfeat_img[n][e] (+)= sum_co colsum[n][co] * wext[co][e]
 *   dlin_w[e][k] = sum_g dfeat[g][e] * agg[g][k], dlin_b[e] = sum_g dfeat[g][e]   (dfeat[g] = sum_{n in g} dfeat_img[n])
 *   dstat[n][k]  = (sum_e dfeat[g(n)][e] * lin_w[e][k]) / |episode g(n)|                                        */
int pemp_cm_bias_bwd_f32(const float* colsum, const float* feat, const int32_t* group, const float* wext, int ldw,
                         float* dwext, int lddw, float* dfeat_img, int accumulate, int N, int Cout, void* stream);
int pemp_cm_linear_bwd_f32(const float* dfeat_img, const int32_t* group, const float* agg, const float* lin_w,
                           float* dlin_w, float* dlin_b, float* dstat, int N, int G, int C2, void* stream);

/* Backward of the comm statistics through loss.backward() in stage-2 training (entry/pemp_stage2.py:78;
 * networks/backbones.py:209-215): given dstat [N][2][C] (gradients of the per-image mean and max),
 *   dx[n][i][c] += mask[n][i] * (dstat[n][0][c] / HW + [i == argmax_i x*mask] * dstat[n][1][c])
 * with mask the POOLED mask that cm_reduce returned (mask_out) and the first maximal index taking the
 * max gradient (torch.max semantics on the host).  Accumulates into dx [N][HW][ldd].              */
int pemp_cm_bwd_add_f32(const float* x, int ldx, const float* mask, const float* dstat, float* dx, int ldd,
                        int N, int HW, int C, void* stream);
/* Training forms: pemp_cm_reduce_arg_f32 also records argmax [N][C] (int32: the first maximal pixel of x*mask per
 * image and channel; NULL = pemp_cm_reduce_f32), and pemp_cm_bwd_add_arg_f32 applies the same gradient from that
 * record as one element-wise pass (no second reduction over x).  C % 4 == 0.                                    */
int pemp_cm_reduce_arg_f32(const float* x, int ldx, const float* mask_in, float* mask_out, float* stat,
                           int32_t* argmax, int N, int Hm, int Wm, int Hx, int Wx, int C, int stride, void* stream);
int pemp_cm_bwd_add_arg_f32(const float* mask, const float* dstat, const int32_t* argmax, float* dx, int ldd,
                            int N, int HW, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training path (Trainer.train_step, entry/pemp_stage1.py:57-65; model.train() at
 * core/base_trainer.py:189).
 * ---------------------------------------------------------------------------------------------- */

/* The conv in front of a train-mode BatchNorm with the BatchNorm's batch statistics started in its epilogue
 * (networks/backbones.py:48-52,66-75 with the model in train(), core/base_trainer.py:189): y = conv(x, w) (no affine,
 * no residual, no ReLU) and, for every group of 32 consecutive output rows r, the per-channel partial sums
 *   stats[r][0][c] = sum y[m][c],  stats[r][1][c] = sum y[m][c]^2        (m in the rows of row tile r, m < M)
 * i.e. pemp_conv2d_stats_rows(d) x 2 x Cout floats (one row per row tile of the chosen variant: ceil(M / 64 .. 256)).
 * pemp_bn_stats_partials_f32 adds them in a fixed order (double) into mean / 1/sqrt(var+eps) and the running-statistics
 * update: the separate pass over y that pemp_bn_stats_f32 makes is gone.  pemp_bn_fwd_partials_f32 = that followed by the
 * normalisation (pemp_bn_apply_mask_f32), one call for the pair.
 * Deterministic.  Non-stem convs whose operands lie below 2 GiB (the buffer-addressed kernels); returns
 * -2 where that does not hold (use pemp_conv2d_nhwc_f32 + pemp_bn_stats_f32 then).  d->tile: 0 or 21..27, or 31..37
 * (not 33) = the same tile shapes with the LAST, partly filled round of tiles split along K (8 images of 51 x 51 pixels
 * give 326 tiles of 128 x 128 for 256 output channels: 256 CUs are busy for two rounds and do the work of 1.27).  The
 * T mod 256 remainder tiles are computed by up to 256 blocks, each over a slice of the K steps; partial accumulators go
 * to `ws`, the block that arrives last (an arrival counter per tile) adds the slices in ascending order -- the result
 * does not depend on arrival order, but differs from the unsplit variants by the rounding of that regrouped sum.
 * ws: pemp_conv2d_splitk_workspace_bytes(d) bytes of UNCACHED device memory from pemp_uncached_alloc (the blocks of one
 * tile run on different XCDs, whose L2s do not see each other's lines: in uncached memory the exchange needs no cache
 * write-back / invalidate -- with agent-scope fences on ordinary memory the variant loses what it gains); its first 1024
 * bytes (the counters) must be zero on entry and are left zero; not shared by launches that may run concurrently.
 * NULL / 0 for the other tile ids.                                                                                      */
size_t pemp_conv2d_splitk_workspace_bytes(const pemp_conv_desc* d);
/* pemp_conv2d_nhwc_f32 with the split-K tile ids (31..37) allowed; other ids behave as there (ws unused).  Used by the
 * training step and by evaluation steps of one or two episodes (<= 12000 output rows, where the unsplit variants leave most
 * CUs idle); larger evaluation steps keep the variants that are bit-identical to each other.                              */
int pemp_conv2d_splitk_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* scale,
                                const float* shift, const float* residual, void* ws, size_t ws_bytes, void* stream);
/* ... and with a per-channel padding VALUE as well (pemp_conv2d_padv_nhwc_f32 + the split-K tile ids): the dilated ASPPV2
 * branch convs of a one-episode evaluation step.                                                                        */
int pemp_conv2d_padv_splitk_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* scale,
                                     const float* shift, const float* residual, const float* pad_value, void* ws,
                                     size_t ws_bytes, void* stream);
void* pemp_uncached_alloc(size_t bytes);       /* zero-filled; NULL on failure (pemp_last_error) */
int pemp_uncached_free(void* p);
/* One idle wave for `us` microseconds on `stream` (no memory traffic).  Two of them on two streams take `us` when the streams
 * run concurrently and 2 x `us` when the runtime maps both to one hardware queue: the training engine uses it to choose a side
 * stream that really runs beside the main one (HIP deals streams round-robin to a few hardware queues).                    */
int pemp_spin_us(int us, void* stream);
/* Zero the arrival counters (first 1024 bytes) of a split-K workspace on `stream`: the kernels leave them zero, a launch
 * that failed may not have -- the host side calls this before it reports the failure.                                     */
int pemp_splitk_reset(void* ws, void* stream);
int pemp_conv2d_stats_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, float* stats,
                               void* ws, size_t ws_bytes, void* stream);
int pemp_conv2d_stats_rows(const pemp_conv_desc* d);
int pemp_bn_stats_partials_f32(const float* stats, int nrows, int M, int C, float eps, float momentum, float* mean,
                               float* invstd, float* run_mean, float* run_var, void* stream);
int pemp_bn_fwd_partials_f32(const float* z, int ldz, const float* stats, int nrows, int M, int C, float eps, float momentum,
                             const float* gamma, const float* beta, const float* residual, int ldr, float* y, int ldy,
                             int relu, uint32_t* mask, float* mean, float* invstd, float* run_mean, float* run_var,
                             void* stream);

/* The input-gradient conv whose result is the gradient at the OUTPUT of a train-mode BatchNorm(+ReLU) (autograd of
 * BottleNeck.forward, networks/backbones.py:66-75: conv -> bn -> relu chains), with the first half of that BatchNorm's
 * backward in its epilogue:  o = conv(x, w) (+ residual);  g = o where the BatchNorm's ReLU let the value through
 * (bit c % 32 of mask[m][c / 32], from pemp_bn_apply_mask_f32; mask NULL: no ReLU, g = o);  y = g;  and per row tile
 *   stats[r][0][c] = sum g[m][c],   stats[r][1][c] = sum g[m][c] * (z[m][c] - mean[c]) * invstd[c]
 * (z: the BatchNorm's input, d->Cout channels, per-pixel stride ldz).  pemp_bn_bwd_partials_f32 finishes: dbeta / dgamma
 * from the partials (fixed order, double) and dz = gamma*invstd*(g - dbeta/M - xhat*dgamma/M): the separate reduction pass
 * over (dy, y, z) of pemp_bn_bwd_f32 and its second read of y are gone, and g doubles as the residual-branch gradient.
 * Same restrictions, tile ids, workspace and return codes as pemp_conv2d_stats_nhwc_f32; d->ldr = per-pixel stride of
 * residual.                                                                                                            */
int pemp_conv2d_bnbwd_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* w, float* y, const float* residual,
                               const uint32_t* mask, const float* z, int ldz, const float* mean, const float* invstd,
                               float* stats, void* ws, size_t ws_bytes, void* stream);
int pemp_bn_bwd_partials_f32(const float* g, int ldg, const float* z, int ldz, const float* mean, const float* invstd,
                             const float* gamma, const float* stats, int nrows, float* dz, int lddz, float* dgamma,
                             float* dbeta, int M, int C, void* stream);

/* Weight gradient of pemp_conv2d_nhwc_f32 (autograd of nn.Conv2d, same call sites):
 *   dw[co][kh][kw][ci] (+)= sum_m g[m][co] * x[pix(m,kh,kw)][ci]      dw is KRSC with row length d->Kpad
 * `d` describes the FORWARD conv (d->ldy = per-pixel stride of g).  STEM4 needs Kpad % 64 == 0.  d->tile: bits 0..7 = kernel choice
 * (0: library's, 1: first generation, 2 / 3: second generation with 128 x 128 / 64 x 64 tiles where the channel counts allow),
 * bits 8.. = number of blocks to aim for when the pixel rows are split over
 * blocks (0: 768; the partial sums of different splits round differently -- same value for workspace query and launch). */
size_t pemp_conv2d_wgrad_workspace_bytes(const pemp_conv_desc* d);
int pemp_conv2d_wgrad_nhwc_f32(const pemp_conv_desc* d, const float* x, const float* g, float* dw,
                               int accumulate, void* ws, size_t ws_bytes, void* stream);

/* nn.BatchNorm2d in train mode (networks/backbones.py:48-52; batch statistics, momentum update
 * of the running statistics with the unbiased variance): per-channel mean / 1/sqrt(var+eps).  */
size_t pemp_colsum_workspace_bytes(int M, int C);
int pemp_bn_stats_f32(const float* z, int ldz, int M, int C, float eps, float momentum,
                      float* mean, float* invstd, float* run_mean, float* run_var,
                      void* ws, size_t ws_bytes, void* stream);
/* y = relu?((z-mean)*invstd*gamma + beta (+ residual))  (BottleNeck.forward, backbones.py:66-75) */
int pemp_bn_apply_f32(const float* z, int ldz, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, const float* residual, int ldr,
                      float* y, int ldy, int M, int C, int relu, void* stream);
/* The same, also leaving the sign of y as one bit per value (mask[m][c / 32] bit c % 32 = y[m][c] > 0; M x C/32 words; NULL:
 * none) for the fused backward above.  C in {32, 64, 128, 256, 512, 1024} when mask is given.                          */
int pemp_bn_apply_mask_f32(const float* z, int ldz, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, const float* residual, int ldr,
                      float* y, int ldy, int M, int C, int relu, uint32_t* mask, void* stream);
/* backward of the above: g = dy*(y>0 if relu) [optionally stored to gout = gradient of the residual
 * branch]; dgamma = sum g*xhat, dbeta = sum g, dz = gamma*invstd*(g - dbeta/M - xhat*dgamma/M).     */
int pemp_bn_bwd_f32(const float* dy, int lddy, const float* y, int ldy, const float* z, int ldz,
                    const float* mean, const float* invstd, const float* gamma,
                    float* dz, int lddz, float* gout, int ldg, float* dgamma, float* dbeta,
                    int M, int C, int relu, void* ws, size_t ws_bytes, void* stream);
/* The same with the sign bits of y (pemp_bn_apply_mask_f32) standing in for y where mask is given: y is then not read. */
int pemp_bn_bwd_mask_f32(const float* dy, int lddy, const float* y, int ldy, const uint32_t* mask, const float* z, int ldz,
                    const float* mean, const float* invstd, const float* gamma,
                    float* dz, int lddz, float* gout, int ldg, float* dgamma, float* dbeta,
                    int M, int C, int relu, void* ws, size_t ws_bytes, void* stream);
/* g = (dy (+ add)) * (y>0 if relu); dbias[c] = sum_m g[m][c] (NULL: skip).  Backward of conv bias + ReLU
 * (networks/pemp_stage1.py:74-78, backbones.py:330-357).                                           */
int pemp_relu_bias_bwd_f32(const float* dy, int lddy, const float* y, int ldy, const float* add, int lda,
                           float* g, int ldg, float* dbias, int M, int C, int relu,
                           void* ws, size_t ws_bytes, void* stream);
/* backward of nn.MaxPool2d (first maximum in scan order wins, as ATen).                           */
int pemp_maxpool2d_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int N, int H, int W, int C,
                                int Ho, int Wo, int k, int s, int p, void* stream);
/* dst[n,hs*s,ws*s,:] = src[n,hs,ws,:], zero elsewhere: input gradient of a stride-s 1x1 conv.     */
int pemp_scatter_strided_nhwc_f32(const float* src, float* dst, int N, int H, int W, int Hs, int Ws,
                                  int C, int s, void* stream);
/* The KRSC weights of the input-gradient convs of EVERY conv layer of a flat parameter buffer, in one launch: layer l's
 * forward weight W[Cout][taps][Cin] at params + off becomes D[Cin][taps][Cout] (taps flipped) at mirror + off -- what
 * pemp_conv2d_nhwc_f32 needs to compute dL/dx as a convolution of dL/dy (autograd of nn.Conv2d w.r.t. its input,
 * reference entry/pemp_stage1.py:61).  table: device int32 [L][6] = {off, Cout, taps, Cin, first tile, tiles along Cin},
 * tiles of 32 x 32 (Cout x Cin) per tap; total_tiles = their sum.  Refreshed once per training step.                     */
int pemp_dgrad_mirror_f32(const float* params, float* mirror, const int32_t* table, int L, int total_tiles, void* stream);
/* dst[n][i][c] += v[n][c] / HW : backward of F.adaptive_avg_pool2d(x,(1,1)).                      */
int pemp_gap_bwd_add_nhwc_f32(const float* v, float* dst, int ld, int N, int HW, int C, void* stream);
/* Same as pemp_eval_tail_f32 with per-pixel CE weights (CELossDT, core/losses.py:33-43): stats[b][0] =
 * sum w*ce over valid pixels, stats[b][1] = sum of w over ALL pixels (the reference's denominator).   */
int pemp_eval_tail_weighted_f32(const float* pred, const int64_t* target, const float* weight,
                                uint8_t* pred_out, float* logits_out, double* stats, void* ws,
                                size_t ws_bytes, int B, int h, int w, int Ho, int Wo, void* stream);

/* CELossDT.boundary2weight (core/losses.py:23-31,35-41): boundary of the fg mask (3x3 dilate/erode), exact
 * Euclidean distance transform, weight = exp(-edt/sigma^2) + 1, all on the device (the reference calls scipy
 * on the host every step).  target int64 [B][H][W] -> weight fp32 [B][H][W].                          */
size_t pemp_cedt_workspace_bytes(int B, int H, int W);
int pemp_cedt_weight_f32(const int64_t* target, float* weight, void* ws, size_t ws_bytes,
                         int B, int H, int W, float sigma, void* stream);

/* Backward of the prototype head under mean cross-entropy: reverse of networks/pemp_stage1.py:142-163,
 * 195-261 + core/losses.py:10.  Inputs are the forward's operands and by-products:
 *   fwd_ws  the workspace pemp_mpm_protos_f32 (p > 0) / pemp_masked_avg_pool_f32(full_res=0) (p == 0) left
 *           behind for the same features and masks; protos [B][J][c]; pred [B][2][n];
 *   target int64 [B][Ho][Wo]; stats [B][8] from pemp_eval_tail_f32 (valid-pixel counts).
 * Outputs: dsup [B*S][n][ldd], dqry [B][n][ldd] (feature gradients), dctr [c][2p] (NULL when p == 0). */
/* The last B * 2 * n int32 words of the workspace receive, per (episode, group, query pixel), the prototype row the cosine
 * backward routed the gradient to (group 0 = foreground rows [0, p), group 1 = background rows [p, 2p); first maximum).    */
size_t pemp_head_bwd_workspace_bytes(int B, int S, int n, int c, int p);
int pemp_head_bwd_f32(const float* sup_feat, const float* qry_feat, int ldf, const float* mask,
                      const float* ctr, const void* fwd_ws, const float* protos, const float* pred,
                      const int64_t* target, const float* weight /* NULL or [B][Ho][Wo] */,
                      const double* stats, float* dsup, float* dqry, int ldd,
                      float* dctr, void* ws, size_t ws_bytes, int B, int S, int h, int w, int H, int W,
                      int Ho, int Wo, int c, int p, int map_full_res /* Baseline: fwd_ws from full_res=1 */,
                      float dist_scalar, void* stream);

/* The same backward for an ARBITRARY upstream gradient of the model output (what autograd's loss.backward()
 * hands to `model(sup, msk, qry, out_shape)`, entry/pemp_stage1.py:59-61): dlogits [B][2][Ho][Wo] replaces
 * (pred, target, weight, stats); the adjoint of the bilinear upsample is applied to it directly.             */
int pemp_head_bwd_dlogits_f32(const float* sup_feat, const float* qry_feat, int ldf, const float* mask,
                              const float* ctr, const void* fwd_ws, const float* protos, const float* dlogits,
                              float* dsup, float* dqry, int ldd, float* dctr, void* ws, size_t ws_bytes,
                              int B, int S, int h, int w, int H, int W, int Ho, int Wo, int c, int p,
                              int map_full_res, float dist_scalar, void* stream);

/* nn.utils.clip_grad_norm_(params, max_norm) + SGD(momentum, weight_decay).step() on flat buffers
 * (entry/pemp_stage1.py:63-64, core/solver.py:87-91).  grad_scale multiplies the gradients first
 * (1/world after a SUM all-reduce); max_norm <= 0 disables clipping; grad_norm_out[0] = ||g||_2.  */
size_t pemp_sgd_workspace_bytes(void);
int pemp_sgd_clip_step_f32(float* params, const float* grads, float* momentum_buf, long long n,
                           float max_norm, float lr, float momentum, float weight_decay, int first_step,
                           float grad_scale, int nesterov, float* grad_norm_out, void* ws, size_t ws_bytes,
                           void* stream);
/* clip_grad_norm_(max_norm) + torch.optim.Adam(lr, betas, eps, weight_decay).step() (reference core/solver.py:92-96,
 * tr.opt = adam; ATen's operation order, L2 weight decay, no amsgrad).  exp_avg / exp_avg_sq: the optimizer state, zero before
 * the first update; step: 1-based number of this update.  Workspace: pemp_sgd_workspace_bytes().  lr / betas / eps /
 * weight_decay are doubles because torch.optim.Adam holds them as Python floats and forms 1 - beta, lr / bias1 and
 * sqrt(bias2) in double before narrowing each to float once (float(1 - 0.999) != 1.f - 0.999f by 1.3e-5 relative).        */
int pemp_adam_clip_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                            float max_norm, double lr, double beta1, double beta2, double eps, double weight_decay,
                            long long step, float grad_scale, float* grad_norm_out, void* ws, size_t ws_bytes, void* stream);

/* Train-time regularisers.  Random numbers: counter-based Philox4x32-10, element i of stream (seed, offset)
 * (reproducible across launches / graph replays; the reference's torch generator stream is not reproducible,
 * so the DRAWS are parity-unpinned).  `uniforms` != NULL substitutes caller-provided U[0,1) numbers -- that is
 * how the tests check the arithmetic exactly.
 *
 * DropBlock2D.forward (dropblock==0.3.0; call sites networks/pemp_stage1.py:76,79; backbones.py:329-353):
 *   seed map (u < drop_prob / block_size^2) per (image, y, x), dilated by max_pool2d(block_size, 1, block_size/2)
 *   (cropped for even sizes); mask = 1 - dilated [N][H][W]; kept_count[0] = sum(mask) (int, exact).           */
/*   step (device pointer or NULL): the stream offset becomes offset + (*step << 20), so a captured hipGraph draws
 *   fresh numbers on every replay once the caller increments *step.                                          */
int pemp_dropblock_mask_f32(float* mask, int* kept_count, const float* uniforms, int N, int H, int W,
                            float drop_prob, int block_size, uint64_t seed, uint64_t offset,
                            const uint64_t* step, void* stream);
/*   y[m][c] = ((x[m][c] * mask[m]) * M) / kept_count   (the layer's forward AND its backward)               */
int pemp_pixel_scale_f32(const float* x, int ldx, const float* mask, const int* kept_count, float* y, int ldy,
                         long long M, int C, void* stream);
/* nn.Dropout2d(p) in train() (networks/pemp_stage2.py:67,70; backbones.py:284-305):
 *   mask[n][c] = (u < 1 - p) / (1 - p);   y[n][i][c] = x[n][i][c] * mask[n][c]  (forward and backward)      */
int pemp_dropout2d_mask_f32(float* mask, const float* uniforms, int N, int C, float p, uint64_t seed,
                            uint64_t offset, const uint64_t* step, void* stream);
int pemp_channel_scale_f32(const float* x, int ldx, const float* mask, float* y, int ldy, int N, int HW, int C,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Episode input pipeline on the device: everything
 * PascalVOCTrain._get_episode (data_kits/pascal_voc.py:185-237) does AFTER the JPEG/PNG decode, on
 * uint8 arrays uploaded as one blob:
 *   F.resize(img, BILINEAR) :144   Pillow ImagingResample, 8-bit fixed point (22 fraction bits),
 *                                  horizontal pass then vertical pass, each rounded to uint8 -- bit-exact;
 *   ColorJitter(.4,.4,.4)   :146   ImageEnhance Brightness/Contrast/Color = Image.blend with a black /
 *                                  mean-gray / gray image, in the drawn order -- bit-exact for given factors;
 *   F.hflip                 :143   ; crop_obj window :26-84 (origin chosen by the host);
 *   ToTensor + Normalize    :141-142  ((x / 255) - mean) / std in fp32 -- bit-exact;
 *   F.resize(label, NEAREST):145   Pillow ImagingScaleAffine (source coordinate accumulated in double);
 *   mask // 255 -> stack(fg, 1 - fg) :209-210 (support) / int64 label :231 (query).
 * One descriptor per decoded sample.  The host fills the first block of fields, calls
 * pemp_episode_plan() (fills the "planned" fields, returns the workspace size), uploads descriptors and
 * pixels, then calls pemp_episode_preprocess().                                                    */
typedef struct pemp_sample_desc {
    int64_t img_off;       /* byte offset of the HWC uint8 RGB source inside blob; < 0: no image        */
    int64_t msk_off;       /* byte offset of the HW uint8 label source ({0,255}); < 0: no label         */
    int64_t img_out;       /* element offset in img_out of this sample's [3][H][W] fp32 result          */
    int64_t msk_out;       /* element offset in planes_out (mask_mode 1) / label_out (modes 2, 3)       */
    int32_t hs, ws;        /* source height, width                                                      */
    int32_t sh, sw;        /* size after F.resize (eval: H, W; train: int(H*f), int(W*f), f in [1,1.5])   */
    int32_t oy, ox;        /* crop window origin inside the resized (and flipped) image; window = H x W */
    int32_t flip;          /* 1: horizontal flip after resize / jitter                                  */
    int32_t jitter_order;  /* 0: no jitter; else stage0 | stage1<<2 | stage2<<4 with 1 = brightness,
                              2 = contrast, 3 = saturation                                              */
    float jitter[3];       /* enhancement factors (brightness, contrast, saturation)                    */
    int32_t mask_mode;     /* 0 none; 1 support planes fp32 [2][H][W]; 2 label int64 [H][W] (resized,
                              flipped, cropped); 3 label int64 [hs][ws] (query at test time: not resized) */
    /* planned (pemp_episode_plan): */
    int32_t ksx, ksy;      /* filter taps per output pixel, horizontal / vertical                       */
    int64_t ws_off;        /* byte offset of this sample's tables and intermediates in the workspace    */
} pemp_sample_desc;

/* Validates descs[0..n) (host memory), fills ksx/ksy/ws_off; returns the workspace bytes (0 on error). */
size_t pemp_episode_plan(pemp_sample_desc* descs_host, int n, int H, int W);
/* blob, descs_dev, outputs, ws: device memory; descs_host: the same descriptors in host memory (read
 * during the call only).  mean/std: host pointers to 3 floats.                                        */
int pemp_episode_preprocess(const uint8_t* blob, const pemp_sample_desc* descs_host,
                            const pemp_sample_desc* descs_dev, int n, int H, int W,
                            const float* mean, const float* std, float* img_out, float* planes_out,
                            int64_t* label_out, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PEMP_HIP_H */
