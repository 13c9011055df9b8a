"""Training step of the Baseline model (VGG-16 or ResNet-50 + projection) on MI355X: counterpart of
``Trainer.train_step`` in the reference's entry/baseline.py:54-62 (no gradient clipping there) for
``networks/baseline.py``.  Reuses the flat-buffer machinery, the conv/BN forward-backward helpers and
the optimizer of ``pemp_amd.train_engine``; only the trunk (VGG chain) and the head (masked average
pooling over bilinearly up-sampled features, done through its adjoint) differ.
"""
import torch

from . import ops, train_ops as T
from .train_engine import Stage1TrainEngine, Stage1Trainer, _Conv, conv2d


class _ResNetProjectionEngine(Stage1TrainEngine):
    """ResNet trunk of stage 1 + ``encoder.projection`` (1x1, 1024 -> out_channels, bias, no ReLU)."""

    def _init_tail(self, model):
        self.proj = _Conv(self.flat, model.encoder.projection)

    def _tail_forward(self, x, tape):
        tape["proj_in"] = x
        return conv2d(x, self.proj.fwd_params(relu=False))

    def _tail_backward(self, dfeat, up=None):
        x = self.tape["proj_in"]
        g = torch.empty_like(dfeat)
        T.relu_bias_bwd(dfeat, None, g, relu=False, ws_cache=self.ws, out=self.proj.conv.bias.grad)
        self.proj.wgrad(x, g, self.ws)
        return conv2d(g, self.proj.dgrad_params())


class _VGGEngine(Stage1TrainEngine):
    """VGG16 chain: 3x3 conv + bias (+ ReLU), four max pools (reference networks/backbones.py:372-405)."""

    def _init_trunk(self, bb):
        from .networks.backbones import VGG_LAYOUT
        self.steps = []
        for item in VGG_LAYOUT:
            if isinstance(item, tuple):
                idx, _, _, _, relu = item
                self.steps.append(("conv", _Conv(self.flat, bb.features[idx], stem=(idx == 0)), relu or bb.last_relu))
            else:
                self.steps.append(("pool", item, None))

    def _init_tail(self, model):
        pass

    def _trunk_forward(self, images_list, tape):
        x = self._pack(images_list)
        recs = []
        for kind, obj, relu in self.steps:
            if kind == "conv":
                y = conv2d(x, obj.fwd_params(relu=relu))
                recs.append((kind, obj, relu, x, y))
            else:
                y, idx = T.maxpool_idx(x.contiguous(), 3, obj, 1)
                recs.append((kind, obj, None, x.shape[1:3], idx))
            x = y
        tape["vgg"] = recs
        return x

    def _tail_forward(self, x, tape):
        return x

    def _tail_backward(self, dfeat, up=None):
        return dfeat

    def _trunk_backward(self, dx):
        for kind, obj, relu, x, y in reversed(self.tape["vgg"]):
            if kind == "pool":
                dx = T.maxpool_idx_bwd(y, dx.contiguous(), x, 3, obj, 1)
                self.flat.cut()                    # stage boundary (segmented graph capture; no-op otherwise)
                continue
            g = torch.empty_like(y)
            T.relu_bias_bwd(dx, y, g, relu=relu, ws_cache=self.ws, out=obj.conv.bias.grad)
            obj.wgrad(x, g, self.ws)
            dx = conv2d(g, obj.dgrad_params()) if not obj.stem else None


class BaselineTrainer(Stage1Trainer):
    """``train_step`` of the Baseline: forward, CE, backward, SGD step -- no clipping (entry/baseline.py:54-62)."""
    map_full_res = True

    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=5e-4, device=None, loss="ce", sigma=5.0,
                 use_graph=False):
        from .core import losses
        from .networks.baseline import net_ingredient
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.model = model
        model.train()
        eng_cls = _VGGEngine if model.backbone_name == "vgg16" else _ResNetProjectionEngine
        self.eng = eng_cls(model, self.device)
        self.lr, self.momentum, self.wd, self.max_norm = lr, momentum, weight_decay, 0.0
        self.protos, self.dist_scalar = 0, net_ingredient.cfg["dist_scalar"]
        self.last_grad_norm, self.nesterov, self.optimizer = None, False, None
        self.use_graph, self._graphs = use_graph, {}
        self.loss_obj = losses.get({"loss": loss, "sigma": sigma})

    def _head_hip(self, feat, sup_mask, qry_msk, B, S, Q):
        loss, pred, dfeat = self._main_head(feat, sup_mask, qry_msk, B, S, Q)
        self.eng.backward(dfeat)
        return loss, pred

    def _main_head(self, feat, sup_mask, qry_msk, B, S, Q):
        """Full-resolution masked average pooling (adjoint form) -> cosine -> upsample + CE, and its backward
        -> (loss, pred, d loss / d features)."""
        if Q != 1:
            raise ValueError("query must be 1")
        eng, ws = self.eng, self.eng.ws
        H, W = sup_mask.shape[-2:]
        msk = sup_mask.reshape(B * S, 2, H, W).contiguous()
        tgt = qry_msk.reshape(-1, *qry_msk.shape[-2:]).contiguous()
        sup, qry = feat[:B * S], feat[B * S:]
        pro = ops.masked_avg_pool(sup, msk, B, S, full_res=True, ws_cache=ws)
        pred = ops.cosine_proto_max(qry, pro, self.dist_scalar)
        wmap = self.loss_obj.weight_map(tgt)
        _, stats, _ = ops.eval_tail(pred, tgt, ws_cache=ws, weight=wmap)
        loss = stats[:, 0].sum() / stats[:, 1].sum()
        dfeat = torch.empty_like(feat)
        T.head_bwd(sup, qry, msk, None, ws[("map", B, S, sup.shape[1], sup.shape[2], sup.shape[3])], pro, pred, tgt, stats,
                   dfeat, B, S, 0, self.dist_scalar, ws_cache=ws, weight=wmap, map_full_res=True)
        return loss.float(), pred, dfeat

    def forward_backward(self, sup_img, sup_mask, qry_img, qry_msk):
        eng = self.eng
        B, S, ch, H, W = sup_img.shape
        Q = qry_img.shape[1]
        eng.flat.attach_grads()
        eng.flat.grad.zero_()
        feat = self.encode(sup_img, sup_mask, qry_img)
        return self._head_hip(feat, sup_mask, qry_msk, B, S, Q)


class PANetTrainer(BaselineTrainer):
    """``train_step`` of entry/panet.py:103-110: ``(loss + loss_coef * align_loss).backward(); optimizer.step()`` (no
    gradient clipping) on the explicit HIP engines -> (loss, align_loss).  The alignment branch (networks/panet.py:149-190)
    is the PEMP head with the roles swapped -- the query features are the "support" (pooled under the predicted masks),
    the support features the "query" -- so its backward is ``pemp_head_bwd_f32`` once more, on swapped operands."""

    def __init__(self, model, loss_coef=1.0, **kw):
        super().__init__(model, **kw)
        from .networks.panet import net_ingredient
        self.dist_scalar = net_ingredient.cfg["dist_scalar"]
        self.loss_coef = loss_coef
        self.align_ws = {}
        self.last_align_loss = None

    def align_backward(self, feat, al, B, S, coef, dfeat):
        """dfeat += coef * d align_loss / d features (``coef``: python float or 0-dim device tensor)."""
        sup, qry = feat[:B * S], feat[B * S:]
        qs, msk = qry, al["qmask"]
        if S != 1:       # one pseudo-episode per support image: its "support" is a copy of the episode's query features
            qs, msk = qry.repeat_interleave(S, dim=0), al["qmask"].repeat_interleave(S, dim=0)
        n, h, w, c = qs.shape
        pro = ops.masked_avg_pool(qs, msk.contiguous(), n, 1, full_res=False, ws_cache=self.align_ws)   # == al["pro_s"]
        tmp = torch.empty((2 * n, h, w, c), dtype=torch.float32, device=feat.device)
        T.head_bwd(qs, sup, msk.contiguous(), None, self.align_ws[("map", n, 1, h, w, c)], pro, al["pred_s"], al["target"],
                   al["stats"], tmp, n, 1, 0, self.dist_scalar, ws_cache=self.align_ws, map_full_res=False)
        dq = tmp[:n] if S == 1 else tmp[:n].view(B, S, h, w, c).sum(dim=1)
        if isinstance(coef, torch.Tensor) or coef != 1.0:
            dfeat[:B * S].add_(tmp[n:] * coef)
            dfeat[B * S:].add_(dq * coef)
        else:
            dfeat[:B * S].add_(tmp[n:])
            dfeat[B * S:].add_(dq)

    def _head_hip(self, feat, sup_mask, qry_msk, B, S, Q):
        from .networks.panet import align_forward
        loss, pred, dfeat = self._main_head(feat, sup_mask, qry_msk, B, S, Q)
        al = align_forward(feat, pred, sup_mask, B, S, Q, self.dist_scalar, self.align_ws)
        if self.loss_coef != 0.0:
            self.align_backward(feat, al, B, S, float(self.loss_coef), dfeat)
        self.eng.backward(dfeat)
        self.last_align_loss = al["loss"]
        return loss, pred

    def train_step(self, sup_img, sup_mask, qry_img, qry_msk=None):
        loss = super().train_step(sup_img, sup_mask, qry_img, qry_msk)
        return loss, self.last_align_loss
