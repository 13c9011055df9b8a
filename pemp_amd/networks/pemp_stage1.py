"""PEMP stage 1 on MI355X: drop-in for the reference's ``networks/pemp_stage1.py``.

Same module surface (``net_ingredient``, ``ModelClass``, ``PEMPStage1``, ``pretrained_weights``,
``backbone_error``), same constructor / ``forward`` signature and ``state_dict`` keys
(reference: networks/pemp_stage1.py:11-18,21-37,54-109,111-163,264); the arithmetic runs on
libpemp_hip.so through ``pemp_amd.engine`` / ``pemp_amd.ops`` and raises if that library is
missing or the tensors are not on the GPU.
"""
from collections import OrderedDict
from pathlib import Path

import torch
import torch.nn as nn

from .. import engine, ops
from ..config import Ingredient
from . import backbones

net_ingredient = Ingredient("net", save_git_info=False)
pretrained_weights = {
    "vgg16": Path(__file__).parents[2] / "data/vgg16-397923af.pth",
    "resnet50": Path(__file__).parents[2] / "data/resnet50-19c8e357.pth",
    "resnet101": Path(__file__).parents[2] / "data/resnet101-5d3b4d8f.pth",
}
backbone_error = "Not supported backbone '{}'. [vgg16, resnet50, resnet101]"
_RES_LAYERS = {"resnet50": (3, 4, 6), "resnet101": (3, 4, 23)}


@net_ingredient.config
def net_config():
    dist_scalar = 20            # int, factor multiplied to the cosine similarity
    init_channels = 3           # int, input channels of the model
    out_channels = 512          # int, output channels of the feature extractor
    backbone = "resnet50"       # str, [vgg16, resnet50, resnet101]
    protos = 3                  # int, prototypes per class (0: plain masked average pooling)
    drop_rate = 0.1             # float, DropBlock rate of the purifier (train only)
    block_size = 4              # int, DropBlock block size (train only)


@net_ingredient.config_hook
def net_hook(config, command_name, logger):
    if config["net"]["backbone"] not in pretrained_weights:
        raise ValueError(backbone_error.format(config["net"]["backbone"]))
    return {}


def import_torchvision_trunk(trunk, path, resnet=True):
    """Pretrained import rules of the reference: ResNet copies torchvision keys up to the first
    ``layer4.*`` (networks/backbones.py:138-157); VGG copies the first 26 tensors (:412-421)."""
    pre = torch.load(str(path), map_location="cpu")
    cur = trunk.state_dict()
    if resnet:
        for key in pre:
            if key.split(".")[0] in ("layer4", "fc"):
                break
            cur[key] = pre[key]
    else:
        ck, pk = list(cur.keys()), list(pre.keys())
        for i in range(26):
            cur[ck[i]] = pre[pk[i]]
    trunk.load_state_dict(cur)


MAX_PROTOS = 8      # the prototype-head kernels are instantiated for 2 * protos <= 8 and <= 16 rows per pixel (csrc/head.hip)


def check_protos(protos, key):
    """The reference accepts any ``protos`` (networks/pemp_stage1.py:26,104-105; default 3); this build's head kernels are
    instantiated for at most MAX_PROTOS per class.  Fail at model construction, not inside a kernel launch."""
    if not 0 <= int(protos) <= MAX_PROTOS:
        raise ValueError(f"{key} = {protos}: this build supports 0 (plain masked average pooling) .. {MAX_PROTOS} prototypes "
                         f"per class (the head kernels keep 2 * protos <= 16 rows per pixel in registers; MAXJ in "
                         f"pemp_amd/csrc/head.hip)")


class _HeadMixin:
    """Episode head shared by stage 1 / stage 2 / baseline: prototypes -> cosine map -> upsample."""

    def _engine_for(self, device):
        """The inference engine (packed weights + activation arena + captured graphs) of the current LANE.  Lane 0 is
        the default; ``with model.lane(k):`` selects replica k -- same weights, its own arena and graphs -- so that
        several single-episode steps can be in flight on different HIP streams (entry.pemp_stage1.Evaluator(lanes=K))."""
        lane = self.__dict__.get("_lane", 0)
        prec = self.__dict__.get("_precision", "f32")
        key = lane if prec == "f32" else (lane, prec)
        engines = self.__dict__.setdefault("_engines", {})
        eng = engines.get(key)
        if eng is None or eng["device"] != device:
            arena = engine.Arena(device, torch.bfloat16 if prec == "bf16" else torch.float32)
            eng = {"device": device, "arena": arena}
            self._build_engine(eng, arena)
            engines[key] = eng
        if key == 0:
            self.__dict__["_engine"] = eng
        return eng

    def precision(self, prec):
        """Context manager: ``with model.precision("bf16"):`` runs the enclosed inference calls on the bf16-OPERAND variant of the
        encoder (bf16 activations and weights between the fp32 stem and the fp32 head, fp32 accumulation) -- the side figure of
        bench.py that shows what exact fp32 costs; stage-1 ResNet encoders only.  The default, and everything the parity
        tests hold, is "f32"."""
        if prec not in ("f32", "bf16"):
            raise ValueError(f"precision must be 'f32' or 'bf16', got {prec!r}")
        if prec == "bf16" and not (getattr(self, "_bf16_variant", False) and getattr(self, "backbone_name", "") != "vgg16"):
            raise ValueError("the bf16 variant exists for the stage-1 ResNet encoders only")
        model = self

        class _Prec:
            def __enter__(self_inner):
                self_inner.prev = model.__dict__.get("_precision", "f32")
                model.__dict__["_precision"] = prec

            def __exit__(self_inner, *exc):
                model.__dict__["_precision"] = self_inner.prev
                return False
        return _Prec()

    def lane(self, k):
        """Context manager: run the enclosed inference calls on engine replica ``k``."""
        model = self

        class _Lane:
            def __enter__(self_inner):
                self_inner.prev = (model.__dict__.get("_lane", 0), ops.SK_SCOPE)
                model.__dict__["_lane"] = k
                ops.SK_SCOPE = k                   # lanes run beside each other: each has its own split-K workspace

            def __exit__(self_inner, *exc):
                model.__dict__["_lane"], ops.SK_SCOPE = self_inner.prev
                return False
        return _Lane()

    @staticmethod
    def _require_eval_gpu(model, *tensors):
        if model.training:
            raise RuntimeError("pemp_amd: lowres()/graph replay are inference paths; call model.eval() "
                               "(model(...) itself works in train() mode and is differentiable)")
        for t in tensors:
            if not t.is_cuda:
                raise RuntimeError("pemp_amd: inputs must live on the GPU; there is no CPU fallback")

    def _head(self, eng, feats, sup_mask, B, S, Q, protos, dist_scalar, ret_ind, ctr):
        """feats: NHWC [B*S + B*Q, h, w, c] (supports first).  -> pred [BQ,2,h,w] (+ resp uint8)."""
        if Q != 1:
            raise ValueError("query must be 1 (the reference's broadcasting requires it, pemp_stage1.py:197,257)")
        ws = eng["arena"].ws
        sup, qry = feats[:B * S], feats[B * S:]
        H, W = sup_mask.shape[-2:]
        msk = sup_mask.reshape(B * S, 2, H, W).contiguous()
        if protos > 0:
            pro = ops.mpm_protos(sup, msk, ctr, B, S, protos, ws_cache=ws)
        else:
            pro = ops.masked_avg_pool(sup, msk, B, S, full_res=False, ws_cache=ws)
        self.__dict__["_last_protos"] = pro
        return ops.cosine_proto_max(qry, pro, dist_scalar, want_resp=ret_ind)

    def lowres_graphed(self, *inputs, ret_ind=False, device=None):
        """``lowres`` replayed from a captured hipGraph (one per input signature).

        The ~80 launches of one episode are short (tens of µs), so eager issue is bound by the
        host (ctypes + launch ≈ 5-10 µs each); the graph removes that.  Inputs are copied into
        static buffers; the returned tensors are the graph's static outputs and are overwritten
        by the next call with the same signature.  ``device``: HOST tensors are accepted and go
        straight into the static buffers on that device (one H2D copy each -- the reference's
        ``test_step`` body hands over host tensors, entry/pemp_stage1.py:48-49 -- instead of a
        fresh device tensor plus a device-to-device copy); the compute still has no CPU path.
        """
        if device is None or any(t.is_cuda for t in inputs):
            self._require_eval_gpu(self, *inputs)
            device = inputs[0].device
        elif self.training:
            self._require_eval_gpu(self)
        eng = self._engine_for(torch.device(device))
        key = tuple((tuple(t.shape), t.dtype) for t in inputs) + (ret_ind, ops.EVAL_SPLITK)    # the conv variants are baked in
        # (the bf16 variant has its own engine, hence its own graphs)
        graphs = eng.setdefault("graphs", {})
        entry = graphs.get(key)
        if entry is None:
            static_in = [torch.empty(t.shape, dtype=t.dtype, device=device) for t in inputs]
            for s, t in zip(static_in, inputs):
                s.copy_(t)
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):                       # warm-up: populates arena + workspaces
                    self.lowres(*static_in, ret_ind=ret_ind)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                out = self.lowres(*static_in, ret_ind=ret_ind)
            entry = (graph, static_in, out)
            graphs[key] = entry
        graph, static_in, out = entry
        for s, t in zip(static_in, inputs):
            s.copy_(t, non_blocking=True)
        graph.replay()
        return out

    @staticmethod
    def _finish(pred, resp, out_shape):
        out = ops.upsample_bilinear_ac(pred, out_shape)
        if resp is None:
            return out
        return out, ops.upsample_nearest_u8_i64(resp, out_shape)


class PEMPStage1(_HeadMixin, backbones.BaseModel):
    """Stage 1 of the Prior-Enhanced network with Meta-Prototypes (reference class of the same name)."""
    _bf16_variant = True

    @net_ingredient.capture
    def __init__(self, logger, backbone, init_channels, out_channels, protos, drop_rate, block_size):
        super().__init__()
        if backbone not in pretrained_weights:
            raise ValueError(backbone_error.format(backbone))
        pretrained = pretrained_weights[backbone]
        self.backbone_name = backbone
        self.drop_rate, self.block_size = drop_rate, block_size          # DropBlock of the purifier / ASPPV2 (train only)
        if backbone == "vgg16":
            trunk = backbones.VGG16Params(init_channels, last_relu=False)
            self.encoder = nn.Sequential(OrderedDict([("backbone", trunk)]))
            self.__class__.__name__ = "PEMP_Stage1/VGG16"
        else:
            trunk = backbones.ResNetParams(init_channels, _RES_LAYERS[backbone], freeze_bn=True)
            self.encoder = nn.Sequential(OrderedDict([
                ("backbone", trunk), ("purifier", backbones.purifier_params(out_channels, v2=True))]))
            self.__class__.__name__ = "PEMP_Stage1/Resnet50" if backbone == "resnet50" else "PEMP_stage1/Resnet101"
        if pretrained is not None and Path(pretrained).exists():
            import_torchvision_trunk(trunk, pretrained, resnet=backbone != "vgg16")
        elif logger is not None:
            logger.info(f"           ==> pretrained file {pretrained} not found: backbone left at random init")
        check_protos(protos, "net.protos")
        self.ctr = nn.Parameter(torch.rand(out_channels, protos * 2), requires_grad=True) if protos > 0 else None
        if logger is not None:
            logger.info(f"           ==> Model {self.__class__.__name__} created")

    # -- engine --------------------------------------------------------------------------------
    def _build_engine(self, eng, arena):
        bb = self.encoder.backbone
        if self.backbone_name == "vgg16":
            eng["trunk"] = engine.VGG16Engine(bb, arena)
            eng["purifier"] = None
        else:
            eng["trunk"] = engine.ResNetEngine(bb, arena)
            eng["purifier"] = engine.PurifierEngine(self.encoder.purifier, arena)
        eng["ctr"] = self.ctr.detach().float().contiguous() if self.ctr is not None else None

    def encode(self, *image_groups):
        """One or more [n_i,3,H,W] fp32 device tensors -> NHWC features [sum n_i,h,w,c] (in order)."""
        dev = image_groups[0].device
        eng = self._engine_for(dev)
        n = sum(g.shape[0] for g in image_groups)
        H, W = image_groups[0].shape[-2:]
        x4 = eng["arena"].get("x4", (n, H, W, 4), torch.float32)
        o = 0
        for g in image_groups:
            ops.pack_input(g.contiguous(), out=x4[o:o + g.shape[0]])
            o += g.shape[0]
        f = eng["trunk"].forward(x4)
        return eng["purifier"].forward(f) if eng["purifier"] is not None else f

    # -- API -----------------------------------------------------------------------------------
    @net_ingredient.capture
    def forward(self, sup_img, sup_mask, qry_img, out_shape=None, ret_ind=False, protos=3, dist_scalar=20):
        """Same contract as the reference (pemp_stage1.py:111-163): returns logits [BQ,2,Ho,Wo]
        (+ int64 response map [BQ,Ho,Wo] when ``ret_ind``).  In ``train()`` mode the logits carry a grad_fn:
        ``loss.backward()`` runs the explicit HIP backward and fills ``p.grad`` (entry/pemp_stage1.py:59-61)."""
        if self.training:
            if ret_ind:
                raise ValueError("ret_ind is an inference option (entry/pemp_stage1.py:213-217); call model.eval()")
            return self._train_bridge("stage1", sup_img.device)(sup_img, sup_mask, qry_img, out_shape)
        self._require_eval_gpu(self, sup_img, sup_mask, qry_img)
        pred, resp = self.lowres(sup_img, sup_mask, qry_img, ret_ind, protos, dist_scalar)
        H, W = sup_img.shape[-2:]
        return self._finish(pred, resp, out_shape if out_shape is not None else (H, W))

    def lowres(self, sup_img, sup_mask, qry_img, ret_ind=False, protos=None, dist_scalar=None):
        """Feature-resolution prediction [BQ,2,h,w] (+ uint8 response) -- everything before the
        final F.interpolate; shape-static, so it is what gets captured into a hipGraph."""
        cfg = net_ingredient.cfg
        protos = cfg["protos"] if protos is None else protos
        dist_scalar = cfg["dist_scalar"] if dist_scalar is None else dist_scalar
        if (self.ctr is None) != (protos == 0):
            protos = 0 if self.ctr is None else self.ctr.shape[1] // 2
        B, S, ch, H, W = sup_img.shape
        Q = qry_img.shape[1]
        eng = self._engine_for(sup_img.device)
        feats = self.encode(sup_img.reshape(B * S, ch, H, W), qry_img.reshape(B * Q, ch, H, W))
        self.__dict__["_last_feats"] = feats
        out = self._head(eng, feats, sup_mask, B, S, Q, protos, dist_scalar, ret_ind, eng["ctr"])
        return out if ret_ind else (out, None)


ModelClass = PEMPStage1
