"""``model(sup_img, sup_mask, qry_img[, qry_prior], out_shape)`` in ``train()`` mode as an autograd node.

The reference's Trainer (entry/pemp_stage1.py:57-65, pemp_stage2.py:72-83, baseline.py:54-62) does
``loss = loss_obj(model(...), target); loss.backward(); clip_grad_norm_(model.parameters(), 1.1);
optimizer.step()`` with a stock ``torch.optim`` optimizer.  This bridge makes exactly that code run on
the HIP training path: the forward runs the explicit train-mode engine (batch-statistics BN, DropBlock /
Dropout2d, communication modules) and returns fp32 logits carrying a ``grad_fn``; when autograd calls the
node's backward with dL/dlogits, the explicit backward (``pemp_head_bwd_dlogits_f32`` -> encoder backward)
fills the flat gradient buffer and every ``p.grad`` is (re)attached as a view of it, so
``clip_grad_norm_`` / ``optimizer.step()`` / ``optimizer.zero_grad()`` (either flavour) behave as usual.
Gradients ACCUMULATE across backward calls like autograd's own (``p.grad`` present -> added).

The fused trainers (``pemp_amd.train_engine.Stage1Trainer`` ...: fused CE backward, one clip+SGD kernel,
one all-reduce bucket, optional hipGraph) stay the fast path; this is the drop-in path.
"""
import torch

from . import ops, train_ops as T


class TrainBridge:
    def __init__(self, model, kind, device):
        from .train_engine import Stage1Trainer
        self.model, self.kind = model, kind
        if kind == "stage1":
            self.trainer = Stage1Trainer(model, device=device)
        elif kind == "baseline":
            from .train_baseline import BaselineTrainer
            self.trainer = BaselineTrainer(model, device=device)
        elif kind == "panet":
            from .train_baseline import PANetTrainer
            self.trainer = PANetTrainer(model, device=device)
        elif kind == "stage2":
            from .train_stage2 import Stage2Trainer
            self.trainer = Stage2Trainer(None, model, device=device)
        else:
            raise ValueError(kind)
        self.anchor = torch.zeros(1, device=device, requires_grad=True)      # makes autograd record the node

    def __call__(self, sup_img, sup_mask, qry_img, out_shape=None, qry_prior=None):
        for t in (sup_img, sup_mask, qry_img):
            if not t.is_cuda:
                raise RuntimeError("pemp_amd: inputs must live on the GPU; there is no CPU fallback")
        if out_shape is None:
            out_shape = tuple(sup_img.shape[-2:])
        # kind "panet": the node has two differentiable outputs, (logits, align_loss) (networks/panet.py:118)
        return _TrainStep.apply(self, self.anchor, sup_img, sup_mask, qry_img, qry_prior, tuple(int(v) for v in out_shape))


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bridge, anchor, sup_img, sup_mask, qry_img, qry_prior, out_shape):
        tr = bridge.trainer
        eng, ws = tr.eng, tr.eng.ws
        B, S, ch, H, W = sup_img.shape
        Q = qry_img.shape[1]
        if Q != 1:
            raise ValueError("query must be 1 (the reference's broadcasting requires it, pemp_stage1.py:197,257)")
        feat = tr.encode(sup_img.float(), sup_mask.float(), qry_img.float(), qry_prior)
        msk = sup_mask.reshape(B * S, 2, H, W).float().contiguous()
        sup, qry = feat[:B * S], feat[B * S:]
        ctr = getattr(tr.model, "ctr", None)
        if tr.protos > 0:
            key = ("mpm", B, S, sup.shape[1], sup.shape[2], sup.shape[3], tr.protos)
            pro = ops.mpm_protos(sup, msk, ctr.data, B, S, tr.protos, ws_cache=ws)
        else:
            key = ("map", B, S, sup.shape[1], sup.shape[2], sup.shape[3])
            pro = ops.masked_avg_pool(sup, msk, B, S, full_res=tr.map_full_res, ws_cache=ws)
        pred = ops.cosine_proto_max(qry, pro, tr.dist_scalar)
        ctx.bridge, ctx.state, ctx.align = bridge, (feat, msk, pro, key, B, S), None
        if bridge.kind == "panet":
            from .networks.panet import align_forward
            ctx.align = align_forward(feat, pred, sup_mask.float(), B, S, Q, tr.dist_scalar, tr.align_ws)
            return ops.upsample_bilinear_ac(pred, out_shape), ctx.align["loss"].clone()
        return ops.upsample_bilinear_ac(pred, out_shape)

    @staticmethod
    def backward(ctx, dlogits, dalign=None):
        tr = ctx.bridge.trainer
        eng, ws, flat = tr.eng, tr.eng.ws, tr.eng.flat
        feat, msk, pro, key, B, S = ctx.state
        ctx.state = None
        if eng.tape is None:
            raise RuntimeError("pemp_amd: backward through a train-mode forward whose activations were already consumed "
                               "(one backward per forward)")
        prev = flat.gather_grads()                     # None unless gradients are being accumulated
        flat.grad.zero_()
        ctr = getattr(tr.model, "ctr", None)
        sup, qry = feat[:B * S], feat[B * S:]
        dfeat = torch.empty_like(feat)
        dctr = T.head_bwd_dlogits(sup, qry, msk, ctr.data if ctr is not None else None, ws[key], pro,
                                  dlogits.float().contiguous(), dfeat, B, S, tr.protos, tr.dist_scalar, ws_cache=ws,
                                  map_full_res=tr.map_full_res)
        if ctx.align is not None and dalign is not None:      # + dL/d(align_loss) * d(align_loss)/d(features)
            tr.align_backward(feat, ctx.align, B, S, dalign.float(), dfeat)
        ctx.align = None
        flat.attach_grads()
        if ctr is not None:
            ctr.grad.copy_(dctr)
        eng.backward(dfeat)
        if prev is not None:
            flat.grad.add_(prev)
        return None, None, None, None, None, None, None
