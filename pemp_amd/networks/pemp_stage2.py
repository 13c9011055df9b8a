"""PEMP stage 2 (prior-enhanced, communication modules) on MI355X: drop-in for the reference's
``networks/pemp_stage2.py`` (module surface :9-19,236; constructor :40-102; forward :104-162)."""
from collections import OrderedDict
from pathlib import Path

import torch
import torch.nn as nn

from .. import engine, ops
from . import backbones
from .pemp_stage1 import (net_ingredient, PEMPStage1, pretrained_weights, backbone_error, check_protos,  # noqa: F401
                          _HeadMixin, _RES_LAYERS)

PriorNet = PEMPStage1


@net_ingredient.config
def priornet_config():
    backbone2 = "resnet50"      # str, feature extractor of stage 2 [resnet50, resnet101]
    protos2 = 3                 # int, prototypes per class
    drop_rate2 = 0.5            # float, Dropout2d rate of the purifier (train only)
    cm = True                   # bool, use the communication module


def import_torchvision_trunk_cm(trunk, path, n=2):
    """ResNetCM.init_weights (networks/backbones.py:249-276): torchvision weights with the new input
    channels (4th stem channel, +n channels of every stage's first block) zero-padded."""
    pre = torch.load(str(path), map_location="cpu")
    cur = trunk.state_dict()
    for i, key in enumerate(pre):
        w = pre[key]
        if "layer4" in key:
            break
        if i == 0:
            w = torch.cat((w, torch.zeros((64, 1, 7, 7), dtype=w.dtype)), dim=1)
        elif "downsample.0.weight" in key or "0.conv1.weight" in key:
            w = torch.cat((w, torch.zeros((w.shape[0], n, 1, 1), dtype=w.dtype)), dim=1)
        cur[key] = w
    trunk.load_state_dict(cur)


def import_torchvision_vgg_cm(trunk, path, n=2):
    """VGG16CM.init_weights as written for ``cm = True`` (networks/backbones.py:503-533): the first 26 tensors of the
    torchvision VGG-16 file in order, the stem zero-padded by one input channel, the first conv of layer2..5 (tensors 4, 8,
    14, 20) by the n communication channels."""
    pre = torch.load(str(path), map_location="cpu")
    pre_keys = list(pre.keys())
    cur = trunk.state_dict()
    cur_keys = list(cur.keys())
    for i in range(26):
        w = pre[pre_keys[i]]
        if i == 0:
            w = torch.cat((w, torch.zeros((64, 1, 3, 3), dtype=w.dtype)), dim=1)
        elif i in (4, 8, 14, 20):
            w = torch.cat((w, torch.zeros((w.shape[0], n, 3, 3), dtype=w.dtype)), dim=1)
        cur[cur_keys[i]] = w
    trunk.load_state_dict(cur)


class PEMPStage2(_HeadMixin, backbones.BaseModel):
    @net_ingredient.capture
    def __init__(self, shot, query, logger, backbone, backbone2, init_channels, out_channels, protos2, drop_rate2, cm):
        super().__init__()
        backbone2 = backbone2 or backbone
        if backbone2 not in pretrained_weights:
            raise ValueError(backbone_error.format(backbone2))
        self.spq = shot + query
        self.backbone2_name = backbone2
        self.drop_rate2 = drop_rate2                                    # Dropout2d of the purifier / ASPP (train only)
        pretrained = pretrained_weights[backbone2]
        if backbone2 == "vgg16":
            # VGG16CM (networks/backbones.py:424-533) without a purifier (pemp_stage2.py:48-55).  As shipped the reference
            # cannot get here with a pretrained file -- VGG16CM.init_weights reads ``self.cm``, which is never set (:518) --
            # so only the crash is not reproduced: the import below is the one that method spells out for cm = True.
            trunk = backbones.VGG16CMParams(init_channels + 1, last_relu=False, shot_query=self.spq)
            self.encoder = nn.Sequential(OrderedDict([("backbone", trunk)]))
            self.__class__.__name__ = "PEMP_Stage2/VGG16" + cm * "+CM"
            if pretrained is not None and Path(pretrained).exists():
                import_torchvision_vgg_cm(trunk, pretrained)
        else:
            trunk = backbones.ResNetCMParams(init_channels + 1, _RES_LAYERS[backbone2], freeze_bn=True, shot_query=self.spq)
            self.encoder = nn.Sequential(OrderedDict([
                ("backbone", trunk), ("purifier", backbones.purifier_params(out_channels, v2=False))]))
            self.__class__.__name__ = "PEMP_Stage2/Resnet50" + cm * "+CM"
            if pretrained is not None and Path(pretrained).exists():
                import_torchvision_trunk_cm(trunk, pretrained)
        check_protos(protos2, "net.protos2")
        self.ctr = nn.Parameter(torch.rand(out_channels, protos2 * 2), requires_grad=True) if protos2 > 0 else None
        self.adaptive_p = None
        if logger is not None:
            logger.info(f"           ==> Model {self.__class__.__name__} created")

    def _build_engine(self, eng, arena):
        if self.backbone2_name == "vgg16":
            eng["trunk"] = engine.VGG16CMEngine(self.encoder.backbone, arena)
            eng["purifier"] = None
        else:
            eng["trunk"] = engine.ResNetCMEngine(self.encoder.backbone, arena)
            eng["purifier"] = engine.PurifierEngine(self.encoder.purifier, arena)
        eng["ctr"] = self.ctr.detach().float().contiguous() if self.ctr is not None else None

    def lowres(self, sup_img, sup_mask, qry_img, qry_prior, ret_ind=False, protos2=None, dist_scalar=None):
        cfg = net_ingredient.cfg
        dist_scalar = cfg["dist_scalar"] if dist_scalar is None else dist_scalar
        protos = 0 if self.ctr is None else self.ctr.shape[1] // 2
        B, S, ch, H, W = sup_img.shape
        Q = qry_img.shape[1]
        if S + Q != self.spq:
            raise ValueError(f"model was built for shot+query={self.spq}, got {S + Q}")
        eng = self._engine_for(sup_img.device)
        a = eng["arena"]
        n = B * (S + Q)
        # priors: support = fg mask, query = stage-1 prediction (pemp_stage2.py:132-135)
        prior = a.get("prior", (n, H, W))
        prior[:B * S].copy_(sup_mask[:, :, 0].reshape(B * S, H, W))
        prior[B * S:].copy_(qry_prior.reshape(B * Q, H, W))
        x4 = a.get("x4", (n, H, W, 4))
        ops.pack_input(sup_img.reshape(B * S, ch, H, W).contiguous(), prior[:B * S], out=x4[:B * S])
        ops.pack_input(qry_img.reshape(B * Q, ch, H, W).contiguous(), prior[B * S:], out=x4[B * S:])
        trunk = eng["trunk"]
        key = ("group", B, S, Q)
        if key not in a.ws:
            g = torch.cat((torch.arange(B).repeat_interleave(S), torch.arange(B).repeat_interleave(Q)))
            a.ws[key] = g.to(device=sup_img.device, dtype=torch.int32)
        trunk.group, trunk.n_groups = a.ws[key], B
        f = trunk.forward(x4, prior)
        if eng["purifier"] is not None:
            f = eng["purifier"].forward(f)
        self.__dict__["_last_feats"] = f
        out = self._head(eng, f, sup_mask, B, S, Q, protos, dist_scalar, ret_ind, eng["ctr"])
        self.adaptive_p = self.__dict__["_last_protos"].permute(0, 2, 1)      # [B,c,2p], pemp_stage2.py:185
        return out if ret_ind else (out, None)

    def forward(self, sup_img, sup_mask, qry_img, qry_prior, out_shape=None, ret_ind=False):
        """Same contract as the reference (pemp_stage2.py:104-162); differentiable in train()."""
        if self.training:
            if ret_ind:
                raise ValueError("ret_ind is an inference option; call model.eval()")
            if self.backbone2_name == "vgg16":
                raise NotImplementedError("stage 2 on VGG16CM is an inference path here: the reference cannot train it as "
                                          "shipped (PEMPStage2 always builds VGG16CM with a pretrained file, which crashes, "
                                          "networks/backbones.py:518), so there is nothing to be parity-checked against")
            return self._train_bridge("stage2", sup_img.device)(sup_img, sup_mask, qry_img, out_shape, qry_prior)
        self._require_eval_gpu(self, sup_img, sup_mask, qry_img, qry_prior)
        pred, resp = self.lowres(sup_img, sup_mask, qry_img, qry_prior, ret_ind)
        return self._finish(pred, resp, out_shape if out_shape is not None else tuple(sup_img.shape[-2:]))


ModelClass = PEMPStage2
