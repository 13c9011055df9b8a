"""Baseline few-shot segmenter on MI355X: drop-in for the reference's ``networks/baseline.py``
(module surface :11-25,152; constructor :42-67; forward :69-118).

Masked average pooling runs over the bilinearly up-sampled support features in the reference
(baseline.py:100-110, a 329 MB temporary per support image at 401x401); here the same sums are
taken through the adjoint of the interpolation, on the 51x51 features (pemp_masked_avg_pool_f32,
full_res=1)."""
from collections import OrderedDict
from pathlib import Path

import torch.nn as nn

from .. import engine, ops
from ..config import Ingredient
from . import backbones
from .pemp_stage1 import _HeadMixin, import_torchvision_trunk

net_ingredient = Ingredient("net", save_git_info=False)
pretrained_weights = {
    "vgg16": Path(__file__).parents[2] / "data/vgg16-397923af.pth",
    "resnet50": Path(__file__).parents[2] / "data/resnet50-19c8e357.pth",
}
backbone_error = "Not supported backbone '{}'. [vgg16, resnet50]"


@net_ingredient.config
def net_config():
    dist_scalar = 20            # factor multiplied to the cosine similarity
    init_channels = 3           # input channels of the model
    backbone = "vgg16"          # model backbone [vgg16, resnet50]
    out_channels = 512          # output features


class Baseline(_HeadMixin, backbones.BaseModel):
    @net_ingredient.capture
    def __init__(self, logger, backbone, init_channels, out_channels):
        super().__init__()
        if backbone not in pretrained_weights:
            raise ValueError(backbone_error.format(backbone))
        pretrained = pretrained_weights[backbone]
        self.backbone_name = backbone
        if backbone == "vgg16":
            trunk = backbones.VGG16Params(init_channels, last_relu=False)
            self.encoder = nn.Sequential(OrderedDict([("backbone", trunk)]))
            self.__class__.__name__ = "Baseline/VGG16"
        else:
            trunk = backbones.ResNetParams(init_channels, (3, 4, 6), freeze_bn=True)
            self.encoder = nn.Sequential(OrderedDict([
                ("backbone", trunk), ("projection", nn.Conv2d(1024, out_channels, kernel_size=1, stride=1, bias=True))]))
            self.__class__.__name__ = "Baseline/Resnet50"
        if Path(pretrained).exists():
            import_torchvision_trunk(trunk, pretrained, resnet=backbone != "vgg16")
        if logger is not None:
            logger.info(f"           ==> Model {self.__class__.__name__} created")

    def _build_engine(self, eng, arena):
        if self.backbone_name == "vgg16":
            eng["trunk"] = engine.VGG16Engine(self.encoder.backbone, arena)
            eng["proj"] = None
        else:
            eng["trunk"] = engine.ResNetEngine(self.encoder.backbone, arena)
            eng["proj"] = engine.conv_params(self.encoder.projection, None, relu=False)

    def lowres(self, sup_img, sup_mask, qry_img, ret_ind=False, dist_scalar=None):
        dist_scalar = net_ingredient.cfg["dist_scalar"] if dist_scalar is None else dist_scalar
        B, S, ch, H, W = sup_img.shape
        Q = qry_img.shape[1]
        if Q != 1:
            raise ValueError("query must be 1")
        eng = self._engine_for(sup_img.device)
        a = eng["arena"]
        n = B * (S + Q)
        x4 = a.get("x4", (n, H, W, 4))
        ops.pack_input(sup_img.reshape(B * S, ch, H, W).contiguous(), out=x4[:B * S])
        ops.pack_input(qry_img.reshape(B * Q, ch, H, W).contiguous(), out=x4[B * S:])
        f = eng["trunk"].forward(x4)
        if eng["proj"] is not None:
            f = ops.conv2d(f, eng["proj"], out=a.get("feat", tuple(f.shape[:3]) + (eng["proj"].cout,)))
        self.__dict__["_last_feats"] = f
        msk = sup_mask.reshape(B * S, 2, H, W).contiguous()
        pro = ops.masked_avg_pool(f[:B * S], msk, B, S, full_res=True, ws_cache=a.ws)
        self.__dict__["_last_protos"] = pro
        return ops.cosine_proto_max(f[B * S:], pro, dist_scalar), None

    def forward(self, sup_img, sup_mask, qry_img, out_shape=None):
        """Same contract as the reference (baseline.py:69-118): logits [BQ,2,Ho,Wo]; differentiable in train()."""
        if self.training:
            return self._train_bridge("baseline", sup_img.device)(sup_img, sup_mask, qry_img, out_shape)
        self._require_eval_gpu(self, sup_img, sup_mask, qry_img)
        pred, _ = self.lowres(sup_img, sup_mask, qry_img)
        return self._finish(pred, None, out_shape if out_shape is not None else tuple(sup_img.shape[-2:]))


ModelClass = Baseline
