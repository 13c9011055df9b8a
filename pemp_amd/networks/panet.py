"""PANet on MI355X: drop-in for the reference's ``networks/panet.py`` (module surface :11-25,193; constructor
:42-66; forward :68-118; alignLoss :149-190).

PANet is the Baseline's forward -- encoder, masked average pooling over the up-sampled support features (taken
through the adjoint of the interpolation, ``pemp_masked_avg_pool_f32(full_res=1)``), cosine map, bilinear upsample
-- plus the prototype-alignment branch: prototypes pooled from the QUERY features under the predicted masks
(``pemp_argmax_masks_f32`` -> ``pemp_masked_avg_pool_f32(full_res=0)``), matched against the SUPPORT features
(``pemp_cosine_proto_max_f32``), cross-entropy against the support foreground mask (``pemp_eval_tail_f32``).  The
same kernels as the PEMP path, with the roles of support and query swapped.
``forward`` returns ``(logits, align_loss)`` like the reference; in ``train()`` mode both carry a grad_fn
(``pemp_amd.autograd``), so ``(loss + loss_coef * align_loss).backward()`` of entry/panet.py:103-110 runs unmodified.
"""
import torch

from .. import ops
from ..config import Ingredient
from . import baseline as _baseline

net_ingredient = Ingredient("net", save_git_info=False)
pretrained_weights = _baseline.pretrained_weights
backbone_error = _baseline.backbone_error


@net_ingredient.config
def net_config():
    dist_scalar = 20            # factor multiplied to the cosine similarity
    init_channels = 3           # input channels of the model
    backbone = "vgg16"          # model backbone [vgg16, resnet50]
    out_channels = 512          # output features


def align_forward(feats, pred, sup_mask, B, S, Q, dist_scalar, ws_cache):
    """The alignment branch on device (panet.py:149-190).  feats NHWC [B*S + B*Q, h, w, c] (supports first), pred
    [BQ,2,h,w] -> dict with the loss (0-dim fp32 tensor) and what its backward needs."""
    if Q != 1:
        raise ValueError("query must be 1")
    H, W = sup_mask.shape[-2:]
    sup, qry = feats[:B * S], feats[B * S:]
    qmask = ops.argmax_masks(pred)                                               # [B,2,h,w]: fg, bg
    pro = ops.masked_avg_pool(qry, qmask, B, Q, full_res=False, ws_cache=ws_cache)          # [B,2,c]
    pro_s = pro if S == 1 else pro.repeat_interleave(S, dim=0)                    # compute_similarity's expansion
    pred_s = ops.cosine_proto_max(sup, pro_s.contiguous(), dist_scalar)          # [BS,2,h,w]
    target = sup_mask.reshape(B * S, 2, H, W)[:, 0].long().contiguous()          # the support fg mask as labels
    _, stats, _ = ops.eval_tail(pred_s, target, ws_cache=ws_cache)
    loss = (stats[:, 0].sum() / stats[:, 1].sum()).float()
    return dict(loss=loss, qmask=qmask, pro=pro, pro_s=pro_s, pred_s=pred_s, target=target, stats=stats)


class PANet(_baseline.Baseline):
    @net_ingredient.capture
    def __init__(self, logger, backbone, init_channels, out_channels):
        super().__init__(None, backbone, init_channels, out_channels)       # explicit: the parent's own ingredient is not consulted
        self.__class__.__name__ = "PANet/VGG16" if backbone == "vgg16" else "PANet/Resnet50"
        if logger is not None:
            logger.info(f"           ==> Model {self.__class__.__name__} created")

    def forward(self, sup_img, sup_mask, qry_img, out_shape=None):
        """Same contract as the reference (panet.py:68-118): (logits [BQ,2,Ho,Wo], align_loss)."""
        if self.training:
            return self._train_bridge("panet", sup_img.device)(sup_img, sup_mask, qry_img, out_shape)
        self._require_eval_gpu(self, sup_img, sup_mask, qry_img)
        dist_scalar = net_ingredient.cfg["dist_scalar"]
        B, S = sup_img.shape[:2]
        Q = qry_img.shape[1]
        pred, _ = self.lowres(sup_img, sup_mask, qry_img, dist_scalar=dist_scalar)
        out = self._finish(pred, None, out_shape if out_shape is not None else tuple(sup_img.shape[-2:]))
        eng = self._engine_for(sup_img.device)
        ws = eng.setdefault("align_ws", {})
        al = align_forward(self.__dict__["_last_feats"], pred, sup_mask.float(), B, S, Q, dist_scalar, ws)
        return out, al["loss"]


ModelClass = PANet
