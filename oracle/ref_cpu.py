"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain torch fp32 ops, reference op order) of the PEMP prototype-matching hot
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``pemp_amd``) never does and fails loudly without its HIP
library.

Pinning: the reference's own tests hold no golden vector for this arithmetic, so
the restatement is pinned by outputs of the reference itself, produced in the build container by
``tests/golden/make_golden.py`` (imports /root/reference/networks/*.py on CPU) and committed under
``tests/golden/*.npz``; ``tests/test_cpu_suite.py`` (``test_oracle_*``) checks this file against them, bit for bit.

Every function cites the reference lines it restates (paths relative to the reference root).
All functions are functional: weights come in as a ``{state_dict key: tensor}`` mapping using
the reference's key names.  ``TRAIN = False``: eval-mode semantics (BatchNorm uses running
statistics, DropBlock/Dropout are identities); ``TRAIN = True``: ``model.train()`` BatchNorm (batch statistics) for the
training-step fixtures and the training cpu_baseline.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------
TRAIN = False     # module switch: True = model.train() semantics for BatchNorm (batch statistics,
                  # running-stat update with momentum 0.1); DropBlock/Dropout2d stay identities, i.e.
                  # drop_rate = 0, unless DROPBLOCK below supplies the draws (the layers' own random streams
                  # cannot be pinned)

# DropBlock2D of the stage-1 purifier / ASPPV2 (networks/pemp_stage1.py:76,79; backbones.py:329-351).  The layer comes from
# the third-party package dropblock==0.3.0, which is not in /root/reference; its published forward is restated in
# _dropblock below.  ``DROPBLOCK = DropBlock(drop_prob, block_size, uniforms)`` makes the train-mode pass apply it with the
# uniform draws GIVEN per layer (tag = the DropBlock2D module's name, e.g. "encoder.purifier.2", ->
# float tensor [N,H,W] in [0,1), what ``torch.rand`` would have produced inside the layer); None: identity.
DROPBLOCK = None


class DropBlock:
    def __init__(self, drop_prob, block_size, uniforms):
        self.drop_prob, self.block_size, self.uniforms = float(drop_prob), int(block_size), uniforms
        self.used = set()


# nn.Dropout2d of the stage-2 purifier / ASPP (networks/pemp_stage2.py:68,71; backbones.py:282-306), same arrangement:
# ``DROPOUT2D = Dropout2d(p, uniforms)`` with uniforms[tag] = float tensor [N,C] in [0,1) -- the draws behind ATen's
# bernoulli_(1 - p) noise tensor (feature_dropout: noise = (u < 1 - p) / (1 - p), one value per (image, channel)).
DROPOUT2D = None


class Dropout2d:
    def __init__(self, p, uniforms):
        self.p, self.uniforms = float(p), uniforms
        self.used = set()


def _dropout2d(x, tag):
    d = DROPOUT2D
    if d is None or not TRAIN or d.p == 0.0:
        return x
    u = d.uniforms[tag]
    d.used.add(tag)
    if tuple(u.shape) != tuple(x.shape[:2]):
        raise ValueError(f"dropout2d {tag}: draws {tuple(u.shape)} for activation {tuple(x.shape)}")
    noise = (u < 1 - d.p).to(x.dtype) / (1 - d.p)
    return x * noise[:, :, None, None]


def _dropblock(x, tag):
    """dropblock.DropBlock2D.forward (0.3.0) in train(): gamma = drop_prob / block_size^2; seeds = rand(N,H,W) < gamma;
    block mask = 1 - max_pool2d(seeds, block_size, stride 1, padding block_size // 2) (last row/column cut for an even
    block size); out = x * mask * numel(mask) / sum(mask)."""
    d = DROPBLOCK
    if d is None or not TRAIN or d.drop_prob == 0.0:
        return x
    u = d.uniforms[tag]
    d.used.add(tag)
    if tuple(u.shape) != (x.shape[0],) + tuple(x.shape[2:]):
        raise ValueError(f"dropblock {tag}: draws {tuple(u.shape)} for activation {tuple(x.shape)}")
    bs = d.block_size
    seeds = (u < d.drop_prob / bs ** 2).to(x.dtype)
    bm = F.max_pool2d(seeds[:, None], kernel_size=(bs, bs), stride=(1, 1), padding=bs // 2)
    if bs % 2 == 0:
        bm = bm[:, :, :-1, :-1]
    bm = 1 - bm.squeeze(1)
    out = x * bm[:, None]
    return out * bm.numel() / bm.sum()


# Decision switches (tests/test_grad_frozen_gpu.py).  The path has three kinds of discrete decisions: the sign of a ReLU's
# input, the winner of a max-pool window, the prototype that wins a group maximum.  ``SWITCHES = Switches()`` makes a forward
# pass RECORD them by tag; ``SWITCHES = Switches(decisions)`` makes it TAKE them (y = x * mask, pooled = x[winner],
# max = x[winner]): a decision-frozen, smooth function.  Evaluated in float64 under autograd it yields the gradient that a
# backward pass which used the same decisions must reproduce up to rounding -- no arg-max that flips between two
# float32 evaluations (the noise floor of the end-to-end gradient comparison) is left in the comparison.
# ``SWITCHES = None`` (default): plain F.relu / F.max_pool2d / Tensor.max, bit for bit.
SWITCHES = None


class Switches:
    def __init__(self, decisions=None):
        self.replay = decisions is not None
        self.d = {} if decisions is None else decisions
        self.used = set()

    def take(self, tag):
        self.used.add(tag)
        return self.d[tag]


class _MaxResult:
    def __init__(self, values, indices):
        self.values, self.indices = values, indices

    def __getitem__(self, i):
        return (self.values, self.indices)[i]


def _relu(x, tag):
    s = SWITCHES
    if s is None:
        return F.relu(x)
    if s.replay:
        m = s.take(tag)
        if m.shape != x.shape:
            raise ValueError(f"switch {tag}: mask {tuple(m.shape)} for activation {tuple(x.shape)}")
        return x * m.to(x.dtype)
    s.d[tag] = x.detach() > 0
    return F.relu(x)


def _max_pool2d(x, k, stride, pad, tag, ceil_mode=False):
    s = SWITCHES
    if s is None:
        return F.max_pool2d(x, k, stride, pad, ceil_mode=ceil_mode)
    if s.replay:
        idx = s.take(tag)                # as F.max_pool2d(return_indices=True): h * W + w inside the input plane
        return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    y, idx = F.max_pool2d(x, k, stride, pad, ceil_mode=ceil_mode, return_indices=True)
    s.d[tag] = idx
    return y


def _group_max(t, dim, tag):
    s = SWITCHES
    if s is None:
        return t.max(dim=dim)
    if s.replay:
        idx = s.take(tag)
        return _MaxResult(t.gather(dim, idx.unsqueeze(dim)).squeeze(dim), idx)
    mv = t.max(dim=dim)
    s.d[tag] = mv.indices
    return mv


def _bn(x, sd, p):
    """nn.BatchNorm2d (networks/backbones.py:48-52,90,111,328): eval -> running statistics;
    train (core/base_trainer.py:189) -> batch statistics, running stats updated in place."""
    if TRAIN and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"].add_(1)             # the module's own counter (nn.BatchNorm2d.forward in train mode)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], TRAIN, 0.1 if TRAIN else 0.0, BN_EPS)


def _conv(x, sd, p, stride=1, padding=0, dilation=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, dilation)


def bottleneck(x, sd, p, stride, dilation, downsample):
    """BottleNeck.forward, networks/backbones.py:64-77 (stride sits on conv1, :47)."""
    out = _relu(_bn(_conv(x, sd, p + ".conv1", stride=stride), sd, p + ".bn1"), p + ".relu1")
    out = _relu(_bn(_conv(out, sd, p + ".conv2", padding=dilation, dilation=dilation), sd, p + ".bn2"), p + ".relu2")
    out = _bn(_conv(out, sd, p + ".conv3"), sd, p + ".bn3")
    if downsample:
        res = _bn(_conv(x, sd, p + ".downsample.0", stride=stride), sd, p + ".downsample.1")
    else:
        res = x
    return _relu(out + res, p + ".relu3")


_LAYER_CFG = {  # name: (stride, dilation) -- networks/backbones.py:97-99
    "layer1": (1, 1), "layer2": (2, 1), "layer3": (1, 2),
}


def _res_layer(x, sd, p, name, blocks):
    stride, dil = _LAYER_CFG[name]
    # block 0 always has a downsample in layers 1-3 (channel change / stride / dilation, :108)
    x = bottleneck(x, sd, f"{p}.{name}.0", stride, dil, True)
    for i in range(1, blocks):
        x = bottleneck(x, sd, f"{p}.{name}.{i}", 1, dil, False)
    return x


def resnet_stem(x, sd, p):
    """conv1 7x7 s2 p3 -> bn -> relu -> maxpool 3/2/1 ceil_mode (backbones.py:89-92,125)."""
    x = _relu(_bn(_conv(x, sd, p + ".conv1", stride=2, padding=3), sd, p + ".bn1"), p + ".relu")
    return _max_pool2d(x, 3, 2, 1, p + ".maxpool", ceil_mode=True)


def resnet(x, sd, p, layers=(3, 4, 6)):
    """ResNet.forward with ret_features=False (networks/backbones.py:124-136)."""
    x = resnet_stem(x, sd, p)
    for name, n in zip(("layer1", "layer2", "layer3"), layers):
        x = _res_layer(x, sd, p, name, n)
    return x


def _comm(x, mask, sd, lin, stride, spq):
    """ResNetCM.comm (networks/backbones.py:208-222)."""
    mask = F.max_pool2d(mask, 3, stride, 1)
    masked = (x * mask).view(x.shape[0], x.shape[1], -1)
    mean = masked.mean(dim=-1).view(x.shape[0] // spq, spq, -1).mean(dim=1)
    mx = _group_max(masked, -1, lin + ".max")[0].view(x.shape[0] // spq, spq, -1).mean(dim=1)
    feat = F.linear(torch.cat([mean, mx], dim=1), sd[lin + ".weight"], sd[lin + ".bias"])
    feat = feat[:, None, :, None, None].expand(-1, spq, -1, *x.shape[-2:])
    return feat.reshape(x.shape[0], -1, *x.shape[-2:]), mask


def resnet_cm(x, mask, sd, p, spq, layers=(3, 4, 6)):
    """ResNetCM.forward (networks/backbones.py:224-247)."""
    mask = F.max_pool2d(mask, 3, 2, 1)
    x1 = resnet_stem(x, sd, p)
    c, mask = _comm(x1, mask, sd, p + ".linear1", 2, spq)
    x2 = _res_layer(torch.cat([x1, c], 1), sd, p, "layer1", layers[0])
    c, mask = _comm(x2, mask, sd, p + ".linear2", 1, spq)
    x3 = _res_layer(torch.cat([x2, c], 1), sd, p, "layer2", layers[1])
    c, mask = _comm(x3, mask, sd, p + ".linear3", 2, spq)
    return _res_layer(torch.cat([x3, c], 1), sd, p, "layer3", layers[2])


_ASPP_DIL = (None, 0, 6, 12, 18)  # aspp_0 = global branch, aspp_1 = 1x1, then dilations


def aspp_v2(x, sd, p):
    """ASPPV2.forward, BN -> (DropBlock) -> conv -> ReLU per branch (backbones.py:324-369)."""
    g = F.adaptive_avg_pool2d(x, (1, 1))
    g = _relu(_conv(_dropblock(_bn(g, sd, p + ".aspp_0.0"), p + ".aspp_0.1"), sd, p + ".aspp_0.2"), p + ".aspp_0.relu")
    outs = [g.expand(-1, -1, *x.shape[-2:])]
    for i in range(1, 5):
        d = _ASPP_DIL[i]
        outs.append(_relu(_conv(_dropblock(_bn(x, sd, f"{p}.aspp_{i}.0"), f"{p}.aspp_{i}.1"), sd, f"{p}.aspp_{i}.2",
                                padding=d, dilation=max(d, 1)), f"{p}.aspp_{i}.relu"))
    return _conv(torch.cat(outs, 1), sd, p + ".layer6")


def aspp(x, sd, p):
    """ASPP.forward, conv -> ReLU -> (Dropout2d) per branch (backbones.py:279-321)."""
    g = _dropout2d(_relu(_conv(F.adaptive_avg_pool2d(x, (1, 1)), sd, p + ".aspp_0.0"), p + ".aspp_0.relu"), p + ".aspp_0.2")
    outs = [g.expand(-1, -1, *x.shape[-2:])]
    for i in range(1, 5):
        d = _ASPP_DIL[i]
        outs.append(_dropout2d(_relu(_conv(x, sd, f"{p}.aspp_{i}.0", padding=d, dilation=max(d, 1)), f"{p}.aspp_{i}.relu"),
                               f"{p}.aspp_{i}.2"))
    return _conv(torch.cat(outs, 1), sd, p + ".layer6")


def purifier(x, sd, p, v2=True):
    """encoder.purifier (networks/pemp_stage1.py:73-80; pemp_stage2.py:65-72)."""
    x = _relu(_conv(x, sd, p + ".0"), p + ".0.relu")
    x = _dropblock(x, p + ".2") if v2 else _dropout2d(x, p + ".2")
    x = _relu(_conv(x, sd, p + ".3", padding=1), p + ".3.relu")
    x = _dropblock(x, p + ".5") if v2 else _dropout2d(x, p + ".5")
    return aspp_v2(x, sd, p + ".6") if v2 else aspp(x, sd, p + ".6")


_VGG = (  # (conv index in nn.Sequential, dilation, relu) / "P2" pool s2 / "P1" pool s1 -- backbones.py:375-397
    (0, 1, True), (2, 1, True), "P2", (5, 1, True), (7, 1, True), "P2",
    (10, 1, True), (12, 1, True), (14, 1, True), "P2",
    (17, 1, True), (19, 1, True), (21, 1, True), "P1",
    (24, 2, True), (26, 2, True), (28, 2, None),
)


def vgg16(x, sd, p, last_relu=False):
    """VGG16.forward (networks/backbones.py:372-405)."""
    npool = 0
    for item in _VGG:
        if item == "P2" or item == "P1":
            x = _max_pool2d(x, 3, 2 if item == "P2" else 1, 1, f"{p}.pool{npool}")
            npool += 1
        else:
            idx, d, relu = item
            x = _conv(x, sd, f"{p}.features.{idx}", padding=d, dilation=d)
            if relu or (relu is None and last_relu):
                x = _relu(x, f"{p}.features.{idx}.relu")
    return x


_VGG_CM = (  # VGG16CM (networks/backbones.py:424-466): (layer, [conv indices], dilation, pool stride | None)
    ("layer1", (0, 2), 1, 2), ("layer2", (0, 2), 1, 2), ("layer3", (0, 2, 4), 1, 2), ("layer4", (0, 2, 4), 1, 1),
    ("layer5", (0, 2, 4), 2, None),
)


def vgg16_cm(x, mask, sd, p, spq, last_relu=False):
    """VGG16CM.forward (networks/backbones.py:482-500): after each of the first four stages the communication module's
    two channels are concatenated; the mask is pooled inside ``comm`` only (strides 2, 2, 2, 1)."""
    for li, (layer, convs, d, pool) in enumerate(_VGG_CM):
        for k, idx in enumerate(convs):
            x = _conv(x, sd, f"{p}.{layer}.{idx}", padding=d, dilation=d)
            if not (pool is None and k == len(convs) - 1 and not last_relu):
                x = F.relu(x)
        if pool is None:
            return x
        x = F.max_pool2d(x, 3, pool, 1)
        c, mask = _comm(x, mask, sd, f"{p}.linear{li + 1}", pool, spq)
        x = torch.cat([x, c], 1)
    return x


def encoder_stage1(x, sd, backbone="resnet50"):
    """PEMPStage1.encoder (networks/pemp_stage1.py:60-100)."""
    if backbone == "vgg16":
        return vgg16(x, sd, "encoder.backbone")
    layers = (3, 4, 6) if backbone == "resnet50" else (3, 4, 23)
    return purifier(resnet(x, sd, "encoder.backbone", layers), sd, "encoder.purifier", v2=True)


def encoder_baseline(x, sd, backbone="vgg16"):
    """Baseline.encoder (networks/baseline.py:48-63)."""
    if backbone == "vgg16":
        return vgg16(x, sd, "encoder.backbone")
    return _conv(resnet(x, sd, "encoder.backbone"), sd, "encoder.projection")


# ------------------------------------------------------------------------------------------
# heads
# ------------------------------------------------------------------------------------------
def compute_similarity(fg_proto, bg_proto, qry_fts, dist_scalar=20):
    """networks/pemp_stage1.py:232-261 (= pemp_stage2.py:204-233 = baseline.py:120-149)."""
    fg = F.cosine_similarity(qry_fts, fg_proto[..., None, None], dim=1) * dist_scalar
    bg = F.cosine_similarity(qry_fts, bg_proto[..., None, None], dim=1) * dist_scalar
    return torch.stack((bg, fg), dim=1)


def mpm(sup_fts, qry_fts, sup_fg, sup_bg, ctr, protos, dist_scalar=20, ret_ind=False):
    """PEMPStage1.mpm (networks/pemp_stage1.py:165-230); returns (pred, response|None, protos[B,c,2p])."""
    B, S, c, h, w = sup_fts.shape
    sup_fts = sup_fts.reshape(-1, c, h * w)
    qry_fts = qry_fts.reshape(-1, c, 1, h, w)
    sup_fg = sup_fg.reshape(-1, 1, h * w)
    sup_bg = sup_bg.reshape(-1, 1, h * w)
    response = None
    if ctr is not None:
        ctr = ctr.view(1, c, protos * 2)
        mask = torch.stack((sup_fg, sup_bg), dim=1)                                   # [BS,2,1,hw]
        D = -((sup_fts.unsqueeze(2) - ctr.unsqueeze(3)) ** 2).sum(dim=1)              # [BS,2p,hw]
        D = D.view(-1, 2, protos, h * w)
        D = (torch.softmax(D, dim=2) * mask).view(-1, 1, protos * 2, h * w)
        masked = sup_fts.view(-1, c, 1, h * w) * D                                    # [BS,c,2p,hw]
        new = (masked.sum(dim=3) / (D.sum(dim=3) + 1e-6)).view(B, S, c, 2, protos)
        new = new.transpose(3, 4).reshape(B, S, c * protos, 2).mean(dim=1)            # [B,cp,2]
        adaptive_p = new.view(B, c, protos, 2).transpose(2, 3).reshape(B, c, -1)      # stage2 :185
        fg_proto, bg_proto = new.view(B, c, protos, 2).unbind(dim=3)                  # [B,c,p]
        mv = _group_max(compute_similarity(fg_proto, bg_proto, qry_fts, dist_scalar), 2, "head.proto_max")
        pred = mv.values
        if ret_ind:
            ind = mv.indices
            response = ind[:, 0].clone()
            sel = pred.argmax(dim=1) == 1
            response[sel] = ind[:, 1][sel] + 3
        return pred, response, adaptive_p
    fg_vecs = torch.sum(sup_fts * sup_fg, dim=-1) / (sup_fg.sum(dim=-1) + 1e-5)
    bg_vecs = torch.sum(sup_fts * sup_bg, dim=-1) / (sup_bg.sum(dim=-1) + 1e-5)
    fg_proto = fg_vecs.view(B, S, c).mean(dim=1)
    bg_proto = bg_vecs.view(B, S, c).mean(dim=1)
    pred = compute_similarity(fg_proto, bg_proto, qry_fts.view(-1, c, h, w), dist_scalar)
    return pred, None, torch.stack((fg_proto, bg_proto), dim=2)


def _finish(pred, response, out_shape):
    out = F.interpolate(pred, out_shape, mode="bilinear", align_corners=True)
    if response is None:
        return out
    r = F.interpolate(response.unsqueeze(1).float(), out_shape, mode="nearest")
    return out, r.squeeze(1).long()


def stage1_forward(sd, sup_img, sup_mask, qry_img, out_shape=None, ret_ind=False,
                   backbone="resnet50", protos=3, dist_scalar=20, ret_lowres=False):
    """PEMPStage1.forward (networks/pemp_stage1.py:111-163)."""
    B, S, ch, H, W = sup_img.shape
    Q = qry_img.shape[1]
    x = torch.cat((sup_img, qry_img), dim=1).view(B * (S + Q), ch, H, W)
    f = encoder_stage1(x, sd, backbone)
    _, c, h, w = f.shape
    f = f.view(B, S + Q, c, h, w)
    m = F.interpolate(sup_mask.view(B * S, 2, H, W), (h, w), mode="nearest")
    fg, bg = m.unbind(dim=1)
    ctr = sd.get("ctr") if protos > 0 else None
    pred, resp, _ = mpm(f[:, :S], f[:, S:], fg, bg, ctr, protos, dist_scalar, ret_ind)
    if ret_lowres:
        return pred, f
    return _finish(pred, resp, out_shape if out_shape is not None else (H, W))


def baseline_forward(sd, sup_img, sup_mask, qry_img, out_shape=None, backbone="vgg16",
                     dist_scalar=20, ret_lowres=False):
    """Baseline.forward (networks/baseline.py:69-118): MAP over bilinearly upsampled features."""
    B, S, C, H, W = sup_img.shape
    Q = qry_img.shape[1]
    x = torch.cat((sup_img, qry_img), dim=1).view(B * (S + Q), C, H, W)
    f = encoder_baseline(x, sd, backbone)
    _, c, h, w = f.shape
    f = f.view(B, S + Q, c, h, w)
    sup = F.interpolate(f[:, :S].reshape(B * S, c, h, w), (H, W), mode="bilinear", align_corners=True)
    qry = f[:, S:].reshape(B * Q, c, h, w)
    mfg, mbg = sup_mask.view(B * S, 2, H, W).split(1, dim=1)
    fgv = torch.sum(sup * mfg, dim=(2, 3)) / (mfg.sum(dim=(2, 3)) + 1e-5)
    bgv = torch.sum(sup * mbg, dim=(2, 3)) / (mbg.sum(dim=(2, 3)) + 1e-5)
    fgp = fgv.view(B, S, -1).mean(dim=1)
    bgp = bgv.view(B, S, -1).mean(dim=1)
    pred = compute_similarity(fgp, bgp, qry, dist_scalar)
    if ret_lowres:
        return pred, f
    return F.interpolate(pred, out_shape if out_shape is not None else (H, W),
                         mode="bilinear", align_corners=True)


def panet_align_loss(qry_fts, pred, sup_fts, sup_mask_fg, Q, dist_scalar=20):
    """PANet.alignLoss (networks/panet.py:149-190): prototypes from the QUERY features under the predicted masks,
    matched against the SUPPORT features, cross-entropy against the support foreground mask."""
    B = qry_fts.size(0) // Q
    c = qry_fts.size(1)
    pm = pred.argmax(dim=1, keepdim=True)
    mfg, mbg = (pm == 1).float(), (pm == 0).float()
    fgp = torch.sum(qry_fts * mfg, dim=(2, 3)) / (mfg.sum((2, 3)) + 1e-5)
    bgp = torch.sum(qry_fts * mbg, dim=(2, 3)) / (mbg.sum((2, 3)) + 1e-5)
    fgp = fgp.view(B, Q, c).mean(dim=1)
    bgp = bgp.view(B, Q, c).mean(dim=1)
    S = sup_fts.shape[0] // B
    if S != 1:                                              # compute_similarity's expansion (panet.py:135-139)
        fgp = fgp.view(B, 1, c).expand(-1, S, -1).reshape(B * S, c)
        bgp = bgp.view(B, 1, c).expand(-1, S, -1).reshape(B * S, c)
    ps = compute_similarity(fgp, bgp, sup_fts, dist_scalar)
    out = F.interpolate(ps, sup_mask_fg.shape[-2:], mode="bilinear", align_corners=True)
    return F.cross_entropy(out, sup_mask_fg.squeeze(dim=1).long())


def panet_forward(sd, sup_img, sup_mask, qry_img, out_shape=None, backbone="vgg16", dist_scalar=20):
    """PANet.forward (networks/panet.py:68-118): the Baseline's forward plus the alignment loss -> (logits, align_loss)."""
    B, S, C, H, W = sup_img.shape
    Q = qry_img.shape[1]
    pred, f = baseline_forward(sd, sup_img, sup_mask, qry_img, out_shape, backbone, dist_scalar, ret_lowres=True)
    _, _, c, h, w = f.shape
    sup_fts = f[:, :S].reshape(B * S, c, h, w)
    qry_fts = f[:, S:].reshape(B * Q, c, h, w)
    mfg = sup_mask.view(B * S, 2, H, W)[:, :1]
    out = F.interpolate(pred, out_shape if out_shape is not None else (H, W), mode="bilinear", align_corners=True)
    return out, panet_align_loss(qry_fts, pred, sup_fts, mfg, Q, dist_scalar)


def encoder_stage2(sd, sup_img, sup_mask, qry_img, qry_prior, backbone2="resnet50"):
    """The encoder call of PEMPStage2.forward (networks/pemp_stage2.py:127-140): images with the prior as 4th channel
    (support: its foreground mask, query: the stage-1 prediction) -> features [B*(S+Q),c,h,w], episode-major."""
    B, S, ch, H, W = sup_img.shape
    Q = qry_img.shape[1]
    img = torch.cat((sup_img, qry_img), dim=1).view(B * (S + Q), ch, H, W)
    prior = torch.cat((sup_mask[:, :, :1], qry_prior.view(B, Q, *qry_prior.shape[-3:]).float()), dim=1)
    prior = prior.view(B * (S + Q), 1, H, W)
    if backbone2 == "vgg16":
        return vgg16_cm(torch.cat((img, prior), dim=1), prior, sd, "encoder.backbone", S + Q)
    f = resnet_cm(torch.cat((img, prior), dim=1), prior, sd, "encoder.backbone", S + Q)
    return purifier(f, sd, "encoder.purifier", v2=False)


def stage2_forward(sd, sup_img, sup_mask, qry_img, qry_prior, out_shape=None, ret_ind=False,
                   protos2=3, dist_scalar=20, ret_lowres=False, backbone2="resnet50"):
    """PEMPStage2.forward (networks/pemp_stage2.py:104-162): ResNet-50+CM with the purifier, or VGG16CM (no purifier)."""
    B, S, ch, H, W = sup_img.shape
    Q = qry_img.shape[1]
    f = encoder_stage2(sd, sup_img, sup_mask, qry_img, qry_prior, backbone2)
    _, c, h, w = f.shape
    f = f.view(B, S + Q, c, h, w)
    m = F.interpolate(sup_mask.view(B * S, 2, H, W), (h, w), mode="nearest")
    fg, bg = m.unbind(dim=1)
    ctr = sd.get("ctr") if protos2 > 0 else None
    pred, resp, adaptive_p = mpm(f[:, :S], f[:, S:], fg, bg, ctr, protos2, dist_scalar, ret_ind)
    if ret_lowres:
        return pred, f, adaptive_p
    return _finish(pred, resp, out_shape if out_shape is not None else (H, W))


# ------------------------------------------------------------------------------------------
# harness pieces: loss, test_step, metric
# ------------------------------------------------------------------------------------------
def ce_loss(logits, target):
    """losses.get('ce') = nn.CrossEntropyLoss(ignore_index=255) (core/losses.py:10)."""
    return F.cross_entropy(logits, target, ignore_index=255)


def test_step(forward_fn, inputs, qry_msk):
    """Evaluator.test_step (entry/pemp_stage1.py:48-53): forward at the GT's size, CE, argmax."""
    logits = forward_fn(*inputs, tuple(qry_msk.shape[-2:]))
    tgt = qry_msk.view(-1, *qry_msk.shape[-2:])
    loss = float(ce_loss(logits, tgt))
    return logits.argmax(dim=1).numpy(), loss, logits


class FewShotMetric:
    """core/metrics.py:4-35 (integer tp/fp/fn table, mIoU over val labels, binary IoU)."""

    def __init__(self, classes):
        self.stat = np.zeros((classes + 1, 3))

    def update(self, pred, ref, cls):
        pred = np.asarray(pred, np.uint8)
        ref = np.asarray(ref, np.uint8)
        for i, ci in enumerate(cls):
            p, r = pred[i], ref[i]
            valid = r != 255
            for j, c in enumerate([0, int(ci)]):
                self.stat[c, 0] += int(((p == j) & (r == j) & valid).sum())
                self.stat[c, 1] += int(((p == j) & (r != j) & valid).sum())
                self.stat[c, 2] += int(((p != j) & (r == j) & valid).sum())

    def miou(self, labels, binary=False):
        stat = np.c_[self.stat[0], self.stat[1:].sum(axis=0)].T if binary else self.stat[labels]
        tp, fp, fn = stat.T
        per = tp / (tp + fp + fn)
        return per, per.mean()


def to_torch_sd(np_sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in np_sd.items()}


def cedt_weight(target, sigma=5.0):
    """CELossDT.boundary2weight + the boundary extraction of CELossDT.__call__ (core/losses.py:23-41), with the
    reference's removed ``np.bool`` spelled ``bool``.  target int64 [B,H,W] -> float32 weight [B,H,W]."""
    from scipy.ndimage import distance_transform_edt
    mask = torch.zeros_like(target, dtype=torch.float32)
    mask[target == 1] = 1
    mask = mask.unsqueeze(1)
    kernel = torch.ones(1, 1, 3, 3, dtype=torch.float32)
    conv = F.conv2d(mask, kernel, padding=1)
    dilated = torch.clamp(conv, 0, 1) - mask
    erosion = mask - torch.clamp(conv - 8, 0, 1)
    boundary = (dilated + erosion).squeeze(1)
    bb = np.around(boundary.numpy()).astype(bool)
    edts = np.stack([distance_transform_edt(np.bitwise_not(b)) for b in bb], axis=0)
    return (torch.exp(-torch.from_numpy(edts) / sigma ** 2) + 1).to(torch.float32)


def celoss_dt(logits, target, sigma=5.0):
    """CELossDT.__call__ (core/losses.py:33-43): sum(CE * w) / sum(w), ignore_index 255."""
    loss = F.cross_entropy(logits, target, ignore_index=255, reduction="none")
    w = cedt_weight(target, sigma)
    return (loss * w).sum() / w.sum()


# ------------------------------------------------------------------------------------------
# one training step under autograd (the training cpu_baseline of bench.py; tests/golden/make_f64.py does the same in fp64)
# ------------------------------------------------------------------------------------------
def train_step(sd, sup_img, sup_mask, qry_img, qry_msk, model="stage1", qry_prior=None, lr=1e-3, weight_decay=5e-4,
               max_norm=1.1, dropblock=None, momentum=0.0, buffers=None):
    """Trainer.train_step (entry/pemp_stage1.py:57-65; stage 2: entry/pemp_stage2.py:72-83): forward with the model in
    train() mode (batch-statistics BatchNorm; Dropout2d as identity; DropBlock as identity unless ``dropblock`` =
    DropBlock(...) gives its draws), CrossEntropyLoss(ignore 255),
    backward by autograd, clip_grad_norm_(1.1) (stage 1 only, entry/pemp_stage1.py:63), one SGD step (core/solver.py:87-91;
    torch.optim.SGD's update: d = grad + wd * p; buf = d on a parameter's first step, momentum * buf + d afterwards; p -= lr *
    buf).  ``buffers`` ({}, kept by the caller across steps) holds the momentum buffers; without it every call is a first
    step.  ``sd`` is updated in place (weights and BN running statistics).  Returns (loss, {name: gradient})."""
    global TRAIN, DROPBLOCK
    frozen = lambda k: ("running" in k or "num_batches" in k or k.endswith("backbone.bn1.weight") or k.endswith("backbone.bn1.bias")
                        or ".downsample.1." in k)                      # freeze_bn: stem + downsample BN affines (backbones.py:93-95,113-117)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not frozen(k)}
    old = (TRAIN, DROPBLOCK)
    TRAIN, DROPBLOCK = True, dropblock
    try:
        H, W = qry_msk.shape[-2:]
        if model == "stage1":
            logits = stage1_forward(sd, sup_img, sup_mask, qry_img, (H, W))
        else:
            logits = stage2_forward(sd, sup_img, sup_mask, qry_img, qry_prior, (H, W))
        loss = ce_loss(logits, qry_msk.view(-1, H, W))
        names = list(leaves)
        grads = dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)))
    finally:
        TRAIN, DROPBLOCK = old
    with torch.no_grad():
        live = [g for g in grads.values() if g is not None]
        if model == "stage1" and max_norm > 0:
            # clip_grad_norm_'s own arithmetic (torch/nn/utils/clip_grad.py): float32 norm of the per-tensor float32 norms
            total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in live]), 2.0)
            train_step.last_grad_norm = float(total)             # what clip_grad_norm_ returns (the norm before clipping)
            coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
            for g in live:
                g.mul_(coef)
        for k, g in grads.items():
            if g is not None:
                sd[k].requires_grad_(False)
                d = g.add(sd[k], alpha=weight_decay)                     # torch.optim.SGD's operations, in its order
                if buffers is not None and momentum != 0.0:
                    d = buffers[k].mul_(momentum).add_(d) if k in buffers else buffers.setdefault(k, d.clone())
                sd[k].add_(d, alpha=-lr)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(False)
    return float(loss.detach()), grads


def step_gradients(sd, sup_img, sup_mask, qry_img, qry_msk, model="stage1", backbone="resnet50", qry_prior=None,
                   dtype=torch.float32, decisions=None, dropblock=None, dropout2d=None, probe=None):
    """Loss and d loss / d parameter of ONE train-mode forward (batch-statistics BatchNorm, CE) evaluated in ``dtype`` under
    autograd; ``sd`` is not modified (no update, running statistics untouched).  ``decisions``: see Switches (None: plain
    ReLU / max); ``dropblock`` / ``dropout2d``: the regularisers with given draws (None: identities).
    ``probe``: a tensor R shaped like the encoder's output [B*(S+Q),c,h,w]; the head and the CE are
    replaced by the linear functional  loss = sum(encoder(images) * R), i.e. d loss / d features = R exactly -- the
    encoder's backward pass alone, without the conditioning of the prototype head.
    -> (loss float, {name: gradient}, decision tags used)."""
    global TRAIN, SWITCHES, DROPBLOCK, DROPOUT2D
    frozen = lambda k: ("running" in k or "num_batches" in k or k.endswith("backbone.bn1.weight") or k.endswith("backbone.bn1.bias")
                        or ".downsample.1." in k)
    w = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()) for k, v in sd.items()}
    leaves = {k: v.requires_grad_(True) for k, v in w.items() if v.is_floating_point() and not frozen(k)}
    old = (TRAIN, SWITCHES, DROPBLOCK, DROPOUT2D)
    TRAIN, SWITCHES, DROPBLOCK, DROPOUT2D = True, (None if decisions is None else Switches(decisions)), dropblock, dropout2d
    try:
        H, W = qry_msk.shape[-2:]
        ins = (sup_img.to(dtype), sup_mask.to(dtype), qry_img.to(dtype))
        if probe is not None:
            if model == "stage2":
                f = encoder_stage2(w, *ins, qry_prior.to(dtype), backbone)
            else:
                x = torch.cat((ins[0], ins[2]), dim=1).flatten(0, 1)          # the image order of *_forward
                f = encoder_stage1(x, w, backbone) if model == "stage1" else encoder_baseline(x, w, backbone)
            loss = (f * probe.to(dtype)).sum()
        else:
            if model == "stage1":
                logits = stage1_forward(w, *ins, (H, W), backbone=backbone)
            elif model == "baseline":
                logits = baseline_forward(w, *ins, (H, W), backbone=backbone)
            elif model == "stage2":
                logits = stage2_forward(w, *ins, qry_prior.to(dtype), (H, W), backbone2=backbone)
            else:
                raise ValueError(model)
            loss = ce_loss(logits, qry_msk.view(-1, H, W))
        names = list(leaves)
        grads = dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)))
        used = set() if SWITCHES is None else set(SWITCHES.used)
    finally:
        TRAIN, SWITCHES, DROPBLOCK, DROPOUT2D = old
    return float(loss.detach()), {k: g for k, g in grads.items() if g is not None}, used


def frozen_gradients(sd, sup_img, sup_mask, qry_img, qry_msk, decisions, model="stage1", backbone="resnet50",
                     dtype=torch.float64, dropblock=None, probe=None, qry_prior=None, dropout2d=None):
    """Loss and d loss / d parameter of ONE train-mode forward (batch-statistics BatchNorm, regularisers off, CE) in which
    every discrete decision -- ReLU sign, max-pool winner, winning prototype -- is TAKEN from ``decisions`` (see Switches),
    evaluated in ``dtype`` under autograd.  ``dropblock`` = DropBlock(...) applies the stage-1 DropBlock layers with the
    given draws (their masks depend on the draws only, so they are part of the frozen function).  ``sd`` is not modified.
    -> (loss float, {name: gradient}, tags used)."""
    return step_gradients(sd, sup_img, sup_mask, qry_img, qry_msk, model=model, backbone=backbone, dtype=dtype,
                          decisions=decisions, dropblock=dropblock, probe=probe, qry_prior=qry_prior, dropout2d=dropout2d)
