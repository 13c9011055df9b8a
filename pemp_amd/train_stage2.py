"""Training step of PEMP stage 2 (ResNet-50 + communication modules) on MI355X: counterpart of
``Trainer.train_step`` in the reference's entry/pemp_stage2.py:72-83 for ``networks/pemp_stage2.py``
in ``train()`` mode.

What differs from stage 1 (pemp_amd/train_engine.py), and how it is laid out here:

* the stem sees 4 channels (RGB + prior, pemp_stage2.py:130-138) -- the NHWC4 stem kernel takes the prior
  plane as its 4th channel, nothing is concatenated;
* ``ResNetCM.comm`` (backbones.py:208-222) appends two channels that are constant over space to the input
  of each stage.  A 1x1 conv over a constant channel is a per-image bias, so the first block's conv1 /
  downsample run on the REAL channels (MFMA implicit GEMM) with ``bias[img][co] = sum_e W[co][C+e] feat[img][e]``
  added in the conv epilogue; backward: per-image column sums of dz give the gradients of the two
  extra weight columns and of ``feat``; the Linear(2C -> 2), the episode mean and the mean/max statistics
  are differentiated explicitly (``pemp_cm_bwd_add_f32`` routes the max gradient to the arg-max pixel);
* the block BatchNorms are trainable here (only the stem / downsample BNs are frozen, backbones.py:175-202);
* the purifier uses ``ASPP`` (conv -> ReLU -> Dropout2d, no BN; backbones.py:279-321) and Dropout2d(drop_rate2);
  the channel masks are drawn by the Philox kernels of dropout.hip (the stream itself has no counterpart in the
  reference; with the draws given -- ``eng.draws`` -- the step is checked against the oracle, tests/test_train_gpu.py);
* gradients are clipped only for the VGG variant (entry/pemp_stage2.py:79-81), i.e. never for ResNet-50.
"""
import torch

from . import ops, train_ops as T
from .ops import ConvParams
from .train_engine import Stage1TrainEngine, Stage1Trainer, _BN, _Conv, _SliceWgrad, _enqueue_wgrad, conv2d

_CM_STRIDES = (2, 1, 2)          # backbones.py:230,235,240


class _ConvCM(_Conv):
    """1x1 conv whose last two input channels are the communication channels ([Cout, C+2, 1, 1])."""

    def __init__(self, flat, conv):
        super().__init__(flat, conv)
        if self.kh != 1 or self.kw != 1:
            raise ValueError("communication channels enter 1x1 convs only")
        self.creal = self.cin - 2
        self.wreal = self.wext = None

    def split(self):
        """Once per step: contiguous copy of the real-channel columns, view of the two comm columns."""
        w = self.flat.krsc(self.conv.weight)                       # [Cout, C+2]
        self.wreal = w[:, :self.creal].contiguous()
        self.wext = w[:, self.creal:]
        return self

    def fwd_params(self, relu=False, with_bias=True):
        return ConvParams(self.wreal, None, None, self.creal, self.cout, 1, 1, self.stride, 0, 1, self.creal, False, relu)

    def dgrad_params(self):
        wd = self.flat.dgrad_krsc(self.conv.weight)[:self.creal]     # rows of the transposed weight = real channels
        return ConvParams(wd, None, None, self.cout, self.creal, 1, 1, 1, 0, 1, self.cout, False, False)

    def _wgrad_now(self, x, g, ws):
        dw = torch.empty((self.cout, self.creal), dtype=torch.float32, device=x.device)
        prm = ConvParams(None, None, None, self.creal, self.cout, 1, 1, self.stride, 0, 1, self.creal, False, False)
        T.conv_wgrad(x, g, prm, dw, ws_cache=ws)
        self.flat.krsc_grad(self.conv.weight)[:, :self.creal].copy_(dw)

    def ext_backward(self, colsum, feat, group, dfeat_img, accumulate):
        """colsum [N,Cout] = per-image sums of dz: writes the gradient of the two comm columns and sets / adds to
        dfeat_img [N,2] (the gradient of the broadcast comm features)."""
        T.cm_bias_bwd(colsum, feat, group, self.wext, self.flat.krsc_grad(self.conv.weight)[:, self.creal:], dfeat_img,
                      accumulate)


class Stage2TrainEngine(Stage1TrainEngine):
    def _init_trunk(self, bb):
        f = self.flat
        self.stem = (_Conv(f, bb.conv1, stem=True), _BN(bb.bn1))
        self.blocks, self.stage_first, self.lin = [], [], []
        for name, lin in (("layer1", bb.linear1), ("layer2", bb.linear2), ("layer3", bb.linear3)):
            self.stage_first.append(len(self.blocks))
            self.lin.append(lin)
            for i, blk in enumerate(getattr(bb, name)):
                mk = _ConvCM if i == 0 else _Conv
                self.blocks.append(dict(
                    c1=mk(f, blk.conv1), b1=_BN(blk.bn1), c2=_Conv(f, blk.conv2), b2=_BN(blk.bn2),
                    c3=_Conv(f, blk.conv3), b3=_BN(blk.bn3),
                    ds=(mk(f, blk.downsample[0]), _BN(blk.downsample[1])) if blk.downsample is not None else None))
        self.drop_rate2 = 0.0

    def _init_tail(self, model):
        f, pur = self.flat, model.encoder.purifier
        self.p0, self.p3 = _Conv(f, pur[0]), _Conv(f, pur[3])
        aspp = pur[6]
        self.aspp_conv = [_Conv(f, getattr(aspp, f"aspp_{i}")[0]) for i in range(5)]
        self.l6 = aspp.layer6
        self.midc = self.aspp_conv[0].cout

    # -- trunk --------------------------------------------------------------------------------
    def _trunk_forward(self, images_list, tape, priors=None, group=None, cnt=None):
        """priors: per entry of images_list a [n_i,H,W] fp32 plane; group: int32 [N] episode of each image;
        cnt: number of episodes."""
        n_groups = int(cnt)
        prior = torch.cat([p.reshape(-1, *p.shape[-2:]) for p in priors]).contiguous()
        mask = ops.cm_reduce(None, prior, 2)[0]                               # backbones.py:227
        y, tape["stem"] = self._cbn_fwd(self._pack(images_list, priors), *self.stem, relu=True)
        x, tape["pool_idx"] = T.maxpool_idx(y, 3, 2, 1, ceil_mode=True)
        tape["pool_in"], tape["blocks"], tape["cm"] = y, [], []
        tape["group"] = group
        for bi, b in enumerate(self.blocks):
            if bi in self.stage_first:
                si = self.stage_first.index(bi)
                lin = self.lin[si]
                mask, stat, arg = ops.cm_reduce(x, mask, _CM_STRIDES[si], want_argmax=True)
                agg, feat = ops.cm_linear(stat, group, lin.weight.data, lin.bias.data, n_groups)   # episode mean, Linear(2C->2)
                c1, ds = b["c1"].split(), b["ds"][0].split()
                tape["cm"].append(dict(x=x, mask=mask, agg=agg, feat=feat, arg=arg))
                x, rec = self._block_fwd(x, b, bias_c1=ops.cm_bias(feat, group, c1.wext),
                                         bias_ds=ops.cm_bias(feat, group, ds.wext))
            else:
                x, rec = self._block_fwd(x, b)
            tape["blocks"].append(rec)
        return x

    def _trunk_backward(self, dx):
        tp = self.tape
        group = tp["group"]
        for bi in range(len(self.blocks) - 1, -1, -1):
            b, rec = self.blocks[bi], tp["blocks"][bi]
            # (a stage's first block gets the context-module gradient added to dx below: not final when it leaves the conv)
            up = tp["blocks"][bi - 1]["r3"] if bi > 0 and bi not in self.stage_first else None
            dx = self._block_bwd(dx, b, rec, up=up)
            if bi not in self.stage_first:
                if bi % 2 == 1:
                    self.flat.cut()
                continue
            si = self.stage_first.index(bi)
            cm, lin = tp["cm"][si], self.lin[si]
            dz1, dzd = rec["r1"]["dz"], rec["rd"]["dz"]
            hw_out = float(dz1.shape[1] * dz1.shape[2])
            n, h, w, c = cm["x"].shape
            dfi = torch.empty((n, 2), dtype=torch.float32, device=dx.device)
            b["c1"].ext_backward(ops.global_avgpool(dz1) * hw_out, cm["feat"], group, dfi, accumulate=False)
            b["ds"][0].ext_backward(ops.global_avgpool(dzd) * hw_out, cm["feat"], group, dfi, accumulate=True)
            dstat = T.cm_linear_bwd(dfi, group, cm["agg"], lin.weight.data, lin.weight.grad, lin.bias.grad)
            T.cm_bwd_add(cm["x"], cm["mask"], dstat.view(n, 2, c), dx, argmax=cm["arg"])   # [N,2,C]: d(mean), d(max) per image
            self.flat.cut()                        # stage boundary (segmented graph capture; no-op otherwise)
        dy = T.maxpool_idx_bwd(tp["pool_idx"], dx, tp["pool_in"].shape[1:3], 3, 2, 1)
        self._cbn_bwd(dy, tp["stem"], *self.stem, need_dx=False)

    # -- purifier: conv+ReLU+Dropout2d twice, ASPP (no BN), layer6 ------------------------------
    def _drop(self, y, n, c, layers):
        """nn.Dropout2d(drop_rate2) in train(): one Bernoulli(1-p)/(1-p) multiplier per (image, channel).  ``layers``: the
        reference names of the Dropout2d modules this call stands for, channel ranges in order; ``self.draws``
        ({layer: uniforms [n, channels]}) replaces the Philox stream by given draws (parity tests)."""
        if self.drop_rate2 <= 0.0:
            return y, None
        u = None
        if self.draws is not None:
            u = torch.cat([self.draws[k] for k in layers], dim=1).contiguous()
            if tuple(u.shape) != (n, c) or u.dtype != torch.float32 or u.device != y.device:
                raise ValueError(f"dropout2d draws of {layers}: want float32 {(n, c)} on {y.device}")
        m = T.dropout2d_mask(n, c, self.drop_rate2, self.rng, self.device, uniforms=u)
        return T.channel_scale(y, m), m

    @staticmethod
    def _drop_bwd(dy, m):
        return dy if m is None else T.channel_scale(dy, m)

    def _tail_forward(self, x, tape):
        nimg, h, w, _ = x.shape
        midc = self.midc
        ya = conv2d(x, self.p0.fwd_params(relu=True))
        xa, ma = self._drop(ya, nimg, ya.shape[-1], ("encoder.purifier.2",))
        yb = conv2d(xa, self.p3.fwd_params(relu=True))
        xb, mb = self._drop(yb, nimg, yb.shape[-1], ("encoder.purifier.5",))
        gap = ops.global_avgpool(xb)
        g0 = conv2d(gap.view(nimg, 1, 1, -1), self.aspp_conv[0].fwd_params(relu=True))
        g0d, m0 = self._drop(g0, nimg, midc, ("encoder.purifier.6.aspp_0.2",))
        l6w = self.l6.weight
        w6 = self.flat.krsc(l6w)                                              # [512, 1280]
        w6g = ConvParams(w6[:, :midc].contiguous(), None, self.l6.bias.data, midc, l6w.shape[0], 1, 1, 1, 0, 1, midc, False, False)
        bias6 = conv2d(g0d, w6g)
        cat = self._new(nimg, h, w, 4 * midc)                                 # post-ReLU branch outputs u_i
        for i in range(1, 5):
            conv2d(xb, self.aspp_conv[i].fwd_params(relu=True), out=cat[..., (i - 1) * midc:i * midc])
        # the four branch Dropout2d layers at once
        catd, ms = self._drop(cat, nimg, 4 * midc, tuple(f"encoder.purifier.6.aspp_{i}.2" for i in range(1, 5)))
        w6m = ConvParams(w6[:, midc:].contiguous(), None, None, 4 * midc, l6w.shape[0], 1, 1, 1, 0, 1, 4 * midc, False, False)
        feat = conv2d(catd, w6m, shift_override=bias6.view(nimg, -1), per_image_shift=True)
        tape.update(p0_in=x, ya=ya, ma=ma, xa=xa, yb=yb, mb=mb, xb=xb, gap=gap, g0=g0, m0=m0, g0d=g0d, cat=cat, catd=catd,
                    ms=ms, w6=w6, hw=(nimg, h, w))
        return feat

    def _tail_backward(self, dfeat, up=None):
        tp, midc = self.tape, self.midc
        nimg, h, w = tp["hw"]
        l6w = self.l6.weight
        cout = l6w.shape[0]
        w6 = tp["w6"]
        dw6 = self.flat.krsc_grad(l6w)
        _enqueue_wgrad(self.flat, _SliceWgrad(dw6[:, midc:], 4 * midc, cout), tp["catd"], dfeat, self.ws)      # side stream
        dcat = conv2d(dfeat, ConvParams(T.dgrad_weight(w6[:, midc:].contiguous(), 1, 1), None, None, cout, 4 * midc,
                                            1, 1, 1, 0, 1, cout, False, False))
        s = ops.global_avgpool(dfeat) * float(h * w)                          # per-image column sums [N, 512]
        self.l6.bias.grad.copy_(s.sum(dim=0))
        _enqueue_wgrad(self.flat, _SliceWgrad(dw6[:, :midc], midc, cout), tp["g0d"], s.view(nimg, 1, 1, -1), self.ws)
        dg0 = conv2d(s.view(nimg, 1, 1, -1), ConvParams(T.dgrad_weight(w6[:, :midc].contiguous(), 1, 1), None, None, cout,
                                                            midc, 1, 1, 1, 0, 1, cout, False, False))
        dcat = self._drop_bwd(dcat, tp["ms"])
        dxb = None
        for i in range(1, 5):
            conv = self.aspp_conv[i]
            g = self._new(nimg, h, w, midc)
            sl = slice((i - 1) * midc, i * midc)
            T.relu_bias_bwd(dcat[..., sl], tp["cat"][..., sl], g, relu=True, ws_cache=self.ws, out=conv.conv.bias.grad)
            conv.wgrad(tp["xb"], g, self.ws)
            dxb = conv2d(g, conv.dgrad_params(), residual=dxb)            # branch gradients accumulate in the epilogue
        conv0 = self.aspp_conv[0]
        dg0 = self._drop_bwd(dg0, tp["m0"])
        g = self._new(nimg, 1, 1, midc)
        T.relu_bias_bwd(dg0, tp["g0"], g, relu=True, ws_cache=self.ws, out=conv0.conv.bias.grad)
        conv0.wgrad(tp["gap"].view(nimg, 1, 1, -1), g, self.ws)
        T.gap_bwd_add(conv2d(g, conv0.dgrad_params()).view(nimg, -1), dxb)
        dxb = self._drop_bwd(dxb, tp["mb"])
        g = torch.empty_like(tp["yb"])
        T.relu_bias_bwd(dxb, tp["yb"], g, relu=True, ws_cache=self.ws, out=self.p3.conv.bias.grad)
        self.p3.wgrad(tp["xa"], g, self.ws)
        dxa = conv2d(g, self.p3.dgrad_params())
        dxa = self._drop_bwd(dxa, tp["ma"])
        g = torch.empty_like(tp["ya"])
        T.relu_bias_bwd(dxa, tp["ya"], g, relu=True, ws_cache=self.ws, out=self.p0.conv.bias.grad)
        self.p0.wgrad(tp["p0_in"], g, self.ws)
        return conv2d(g, self.p0.dgrad_params())


class Stage2Trainer(Stage1Trainer):
    """``train_step(*inputs, qry_msk=...)`` of the reference's stage-2 Trainer: frozen stage-1 prior (eval, HIP
    inference path) -> stage-2 forward, CE, backward, SGD (entry/pemp_stage2.py:72-83)."""

    def __init__(self, stage1, model, lr=1e-3, momentum=0.9, weight_decay=5e-4, device=None, drop_rate2=None,
                 loss="ce", sigma=5.0, use_graph=False):
        from .core import losses
        from .networks.pemp_stage2 import net_ingredient
        cfg = net_ingredient.cfg
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stage1, self.model = stage1, model
        if stage1 is not None:
            stage1.eval()
        model.train()
        self.eng = Stage2TrainEngine(model, self.device)
        self.eng.drop_rate2 = getattr(model, "drop_rate2", cfg["drop_rate2"]) if drop_rate2 is None else drop_rate2
        self.lr, self.momentum, self.wd, self.max_norm = lr, momentum, weight_decay, 0.0
        self.protos = 0 if model.ctr is None else model.ctr.shape[1] // 2
        self.dist_scalar = cfg["dist_scalar"]
        self.last_grad_norm, self.nesterov, self.optimizer = None, False, None
        self.use_graph, self._graphs = use_graph, {}
        self.loss_obj = losses.get({"loss": loss, "sigma": sigma})
        self._groups = {}

    def prior(self, sup_img, sup_mask, qry_img):
        """argmax of the stage-1 logits at the input resolution, [BQ,1,H,W] (entry/pemp_stage2.py:74-75)."""
        with torch.no_grad():
            pred, _ = self.stage1.lowres(sup_img, sup_mask, qry_img)
            am, _, _ = ops.eval_tail(pred, None, out_hw=tuple(sup_img.shape[-2:]), ws_cache=self.eng.ws)   # fused upsample+argmax
            return am.unsqueeze(1)

    def encode(self, sup_img, sup_mask, qry_img, qry_prior=None):
        """4-channel input (RGB + prior: support fg mask / query prior, pemp_stage2.py:130-138) -> features."""
        B, S, ch, H, W = sup_img.shape
        Q = qry_img.shape[1]
        if S + Q != self.model.spq:
            raise ValueError(f"model was built for shot+query={self.model.spq}, got {S + Q}")
        if qry_prior is None:
            raise ValueError("stage 2 needs qry_prior (the stage-1 argmax, entry/pemp_stage2.py:74-77)")
        key = (B, S, Q)
        if key not in self._groups:
            g = torch.cat((torch.arange(B).repeat_interleave(S), torch.arange(B).repeat_interleave(Q)))
            self._groups[key] = (g.to(device=self.device, dtype=torch.int32), B)
        priors = [sup_mask[:, :, 0].reshape(B * S, H, W).float(), qry_prior.reshape(B * Q, H, W).float()]
        return self.eng.forward([sup_img.reshape(B * S, ch, H, W), qry_img.reshape(B * Q, ch, H, W)], priors=priors,
                                group=self._groups[key][0], cnt=self._groups[key][1])

    def forward_backward(self, sup_img, sup_mask, qry_img, qry_msk, qry_prior=None):
        B, S = sup_img.shape[:2]
        Q = qry_img.shape[1]
        if qry_prior is None:
            qry_prior = self.prior(sup_img, sup_mask, qry_img)
        self.eng.flat.attach_grads()
        self.eng.flat.grad.zero_()
        feat = self.encode(sup_img, sup_mask, qry_img, qry_prior)
        return self._head_hip(feat, sup_mask, qry_msk, B, S, Q)

    def train_step(self, sup_img, sup_mask, qry_img, qry_msk=None, qry_prior=None):
        ins = [t.to(self.device) for t in (sup_img, sup_mask, qry_img, qry_msk)]
        ins.append(self.prior(*ins[:3]) if qry_prior is None else qry_prior.to(self.device))
        self.eng.buckets.enabled = self.collectives and not self.use_graph   # purifier / ASPP bucket goes out under the trunk's backward
        loss = self._graphed_forward_backward(*ins) if self.use_graph else self.forward_backward(*ins)[0]
        self.optimizer_step()
        self.eng.buckets.enabled = False
        return loss
