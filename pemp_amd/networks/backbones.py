"""Parameter containers of the PEMP encoders + the ``BaseModel`` checkpoint API.

These classes reproduce the reference's ``state_dict`` key layout (networks/backbones.py:42-533;
key list in SURVEY.md §8 a13, pinned by tests/golden/state_keys_*.json) so that reference
checkpoints load unchanged.  They hold parameters only: the arithmetic runs in
``pemp_amd.engine`` on the HIP kernels of libpemp_hip.so, never through ``nn.Conv2d.forward``.
"""
from pathlib import Path

import torch
import torch.nn as nn

__all__ = ["BaseModel", "ResNetParams", "ResNetCMParams", "ASPPParams", "ASPPV2Params", "VGG16Params",
           "purifier_params"]


class BaseModel(nn.Module):
    """Checkpoint helpers with the reference's signatures (networks/backbones.py:21-39)."""

    def load_weights(self, ckpt_path, logger):
        blob = torch.load(str(ckpt_path), map_location="cpu")
        self.load_state_dict(blob["state_dict"] if "state_dict" in blob else blob)
        try:
            shown = Path(ckpt_path).relative_to(Path(__file__).parents[2])
        except ValueError:
            shown = ckpt_path
        logger.info(f"           ==> Model {self.__class__.__name__} initialized from {shown}")

    def maybe_fix_params(self, fix=False):
        if fix:
            for prm in self.parameters():
                prm.requires_grad = False

    # any change of the parameters invalidates the packed device copies used by the engine
    def _invalidate(self, bridge=False):
        self.__dict__["_engine"] = None
        self.__dict__["_engines"] = {}          # every lane's replica (pemp_stage1._HeadMixin._engine_for)
        if bridge:                       # parameters were re-created (device / dtype move): the flat buffers are stale
            self.__dict__["_bridge"] = None

    def _train_bridge(self, kind, device):
        """The train()-mode forward as an autograd node over the explicit HIP training engine (pemp_amd.autograd)."""
        br = self.__dict__.get("_bridge")
        if br is None:
            from ..autograd import TrainBridge
            br = self.__dict__["_bridge"] = TrainBridge(self, kind, device)
        return br

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate(bridge=True)
        return out

    def train(self, mode=True):
        out = super().train(mode)
        self._invalidate()
        return out


def _conv(cin, cout, k, stride=1, pad=0, dil=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=bias)


def _freeze(mod):
    for prm in mod.parameters():
        prm.requires_grad = False


class _Block(nn.Module):
    """conv1/bn1, conv2/bn2, conv3/bn3 (+ downsample.{0,1}) of one bottleneck (backbones.py:42-62)."""

    def __init__(self, cin, planes, stride, dil, with_ds, freeze_ds_bn):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 1, stride), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, 1, dil, dil), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.stride, self.dil = stride, dil
        if with_ds:
            self.downsample = nn.Sequential(_conv(cin, planes * 4, 1, stride), nn.BatchNorm2d(planes * 4))
            if freeze_ds_bn:
                _freeze(self.downsample[1])
        else:
            self.downsample = None


_STAGES = (("layer1", 64, 1, 1), ("layer2", 128, 2, 1), ("layer3", 256, 1, 2))


class ResNetParams(nn.Module):
    """ResNet-50/101 trunk through layer3 (backbones.py:80-122).  ``extra_in`` adds the two
    communication channels in front of each stage for the CM variant (backbones.py:188-206)."""

    def __init__(self, init_c, layers, freeze_bn=True, extra_in=0):
        super().__init__()
        self.conv1, self.bn1 = _conv(init_c, 64, 7, 2, 3), nn.BatchNorm2d(64)
        if freeze_bn:
            _freeze(self.bn1)      # only the stem BN and the downsample BNs end up frozen
        cin = 64
        self.layers = tuple(layers)
        for (name, planes, stride, dil), nblk in zip(_STAGES, layers):
            blocks = [_Block(cin + extra_in, planes, stride, dil, True, freeze_bn)]
            cin = planes * 4
            blocks += [_Block(cin, planes, 1, dil, False, False) for _ in range(1, nblk)]
            setattr(self, name, nn.Sequential(*blocks))


class ResNetCMParams(ResNetParams):
    """ResNetCM: 4-channel stem, +2 channels per stage, three Linear(2c -> 2) (backbones.py:160-206)."""

    def __init__(self, init_c, layers, freeze_bn=True, shot_query=None):
        super().__init__(init_c, layers, freeze_bn, extra_in=2)
        self.spq = shot_query
        self.linear1 = nn.Linear(2 * 64, 2)
        self.linear2 = nn.Linear(2 * 256, 2)
        self.linear3 = nn.Linear(2 * 512, 2)


#: VGG16CM stages (reference networks/backbones.py:431-463): (name, (cin of each conv), cout, dilation, pool stride | None);
#: the first conv of layer2..5 takes the previous stage's channels + the 2 communication channels
VGG_CM_LAYOUT = (("layer1", 2, 64, 1, 2), ("layer2", 2, 128, 1, 2), ("layer3", 3, 256, 1, 2), ("layer4", 3, 512, 1, 1),
                 ("layer5", 3, 512, 2, None))


class VGG16CMParams(nn.Module):
    """VGG16 with communication modules: ``layer{1..5}.{0,2,4}.{weight,bias}``, ``linear{1..4}`` (backbones.py:424-470).
    ReLU / MaxPool positions of the reference's nn.Sequentials are kept as parameter-free placeholders so that the
    state_dict keys are the reference's."""

    def __init__(self, init=4, last_relu=False, shot_query=None):
        super().__init__()
        self.spq, self.last_relu = shot_query, last_relu
        cin = init
        for name, nconv, cout, d, pool in VGG_CM_LAYOUT:
            mods = []
            for k in range(nconv):
                mods += [_conv(cin if k == 0 else cout, cout, 3, 1, d, d, bias=True), nn.Identity()]
            if pool is None and not last_relu:
                mods.pop()
            elif pool is not None:
                mods.append(nn.Identity())
            setattr(self, name, nn.Sequential(*mods))
            cin = cout + 2
        for i, c in enumerate((64, 128, 256, 512), start=1):
            setattr(self, f"linear{i}", nn.Linear(2 * c, 2))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity="relu")


_ASPP_DIL = (0, 0, 6, 12, 18)


class ASPPV2Params(nn.Module):
    """Five branches ``BN -> DropBlock -> conv -> ReLU`` + layer6 (backbones.py:324-358)."""

    def __init__(self, inc=256, midc=256, outc=512):
        super().__init__()
        for i, d in enumerate(_ASPP_DIL):
            k = 1 if i < 2 else 3
            setattr(self, f"aspp_{i}", nn.Sequential(nn.BatchNorm2d(inc), nn.Identity(),
                                                     _conv(inc, midc, k, 1, d, max(d, 1), bias=True), nn.Identity()))
        self.layer6 = _conv(midc * 5, outc, 1, bias=True)


class ASPPParams(nn.Module):
    """Five branches ``conv -> ReLU -> Dropout2d`` + layer6 (backbones.py:279-309)."""

    def __init__(self, inc=256, midc=256, outc=512):
        super().__init__()
        for i, d in enumerate(_ASPP_DIL):
            k = 1 if i < 2 else 3
            setattr(self, f"aspp_{i}", nn.Sequential(_conv(inc, midc, k, 1, d, max(d, 1), bias=True),
                                                     nn.Identity(), nn.Identity()))
        self.layer6 = _conv(midc * 5, outc, 1, bias=True)


def purifier_params(outc, v2):
    """encoder.purifier: indices 0, 3, 6 carry parameters (pemp_stage1.py:73-80; pemp_stage2.py:65-72)."""
    return nn.Sequential(_conv(1024, 256, 1, bias=True), nn.Identity(), nn.Identity(),
                         _conv(256, 256, 3, 1, 1, bias=True), nn.Identity(), nn.Identity(),
                         ASPPV2Params(256, 256, outc) if v2 else ASPPParams(256, 256, outc))


#: VGG16 ``features`` indices: conv (index, cin, cout, dilation, relu) or pool stride (backbones.py:375-397)
VGG_LAYOUT = ((0, 3, 64, 1, True), (2, 64, 64, 1, True), 2,
              (5, 64, 128, 1, True), (7, 128, 128, 1, True), 2,
              (10, 128, 256, 1, True), (12, 256, 256, 1, True), (14, 256, 256, 1, True), 2,
              (17, 256, 512, 1, True), (19, 512, 512, 1, True), (21, 512, 512, 1, True), 1,
              (24, 512, 512, 2, True), (26, 512, 512, 2, True), (28, 512, 512, 2, False))


class VGG16Params(nn.Module):
    """VGG16 with dilated conv5 and stride-1 pool4; ``features.{idx}.{weight,bias}`` (backbones.py:372-399)."""

    def __init__(self, init=3, last_relu=False):
        super().__init__()
        mods = [nn.Identity() for _ in range(29 + (1 if last_relu else 0))]
        for item in VGG_LAYOUT:
            if isinstance(item, tuple):
                idx, cin, cout, d, _ = item
                mods[idx] = _conv(init if idx == 0 else cin, cout, 3, 1, d, d, bias=True)
        self.features = nn.Sequential(*mods)
        self.last_relu = last_relu
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
