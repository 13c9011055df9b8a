"""Execution engine: walks a parameter tree (pemp_amd.networks.backbones), packs it once into
device-resident KRSC weights + folded per-channel affines, and runs the encoder as a chain of
libpemp_hip.so launches on NHWC activations.

Eval-mode arithmetic only in this round (BatchNorm folded with running statistics exactly like
ATen's eval kernel: alpha = weight * rsqrt(var + eps), beta = bias - mean * alpha; DropBlock and
Dropout2d are identities).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ConvParams

BN_EPS = 1e-5


def bn_affine(bn):
    """(alpha, beta) of an eval-mode BatchNorm2d, fp32, on the BN's device."""
    invstd = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    alpha = bn.weight.detach().float() * invstd
    beta = bn.bias.detach().float() - bn.running_mean.detach().float() * alpha
    return alpha.contiguous(), beta.contiguous()


def conv_params(conv, bn=None, relu=False, stem4=False, in_slice=None, dtype=torch.float32):
    """Pack one nn.Conv2d (+ following BN) for pemp_conv2d_nhwc_f32 (``dtype`` bfloat16: the weights of the bf16 side-figure
    variant; the folded affine stays fp32)."""
    w = conv.weight.detach()
    if in_slice is not None:
        w = w[:, in_slice[0]:in_slice[1]]
    packed, kpad = ops.pack_conv_weight(w, stem4=stem4)
    cout, cin = w.shape[0], w.shape[1]
    scale = shift = None
    if bn is not None:
        scale, shift = bn_affine(bn)
        if conv.bias is not None:
            shift = shift + conv.bias.detach().float() * scale
    elif conv.bias is not None:
        shift = conv.bias.detach().float().contiguous()
    return ConvParams(packed.contiguous().to(dtype), scale, shift, 4 if stem4 else cin, cout, conv.kernel_size[0],
                      conv.kernel_size[1], conv.stride[0], conv.padding[0], conv.dilation[0], kpad, stem4, relu)


#: Steps of at most this many feature rows (one or two episodes: 5202 rows per 1-shot episode) issue their INDEPENDENT convs
#: as ONE grouped launch (ops.conv2d_group): the dilated ASPP branches, a stage's downsample conv beside its conv1.  A
#: 5202-row conv is 41-82 tiles on 256 CUs; grouped, the members fill the chip without splitting K and without a launch +
#: drain each.  Same tiles, same K order: results do not change.  PEMP_EVAL_GROUP_ROWS=0 switches it off.
GROUP_MAX_ROWS = int(os.environ.get("PEMP_EVAL_GROUP_ROWS", "12000"))


class Arena:
    """Named activation buffers reused across calls (static addresses make hipGraph replay valid)."""

    def __init__(self, device, dtype=torch.float32):
        self.device = device
        self.dtype = dtype            # activations between the stem and the encoder's last layer (bfloat16: the side-figure variant)
        self.bufs = {}
        self.ws = {}

    def get(self, name, shape, dtype=None, zero=False):
        """``zero``: cleared ONCE, when the buffer is created (for buffers with regions nobody writes afterwards).
        ``dtype`` None: the arena's activation dtype."""
        dtype = self.dtype if dtype is None else dtype
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self.bufs[key] = t
        return t


class _BlockPlan:
    def __init__(self, blk, extra_in=0, dtype=torch.float32):
        # extra_in: the CM variant's two spatially-constant channels are applied as a per-image
        # bias (see ResNetCMEngine), so the packed weights cover only the real channels.
        sl = None
        if extra_in:
            sl = (0, blk.conv1.weight.shape[1] - extra_in)
        self.c1 = conv_params(blk.conv1, blk.bn1, relu=True, in_slice=sl, dtype=dtype)
        self.c2 = conv_params(blk.conv2, blk.bn2, relu=True, dtype=dtype)
        self.c3 = conv_params(blk.conv3, blk.bn3, relu=True, dtype=dtype)     # relu after the residual add
        self.ds = conv_params(blk.downsample[0], blk.downsample[1], relu=False, in_slice=sl, dtype=dtype) \
            if blk.downsample is not None else None
        if extra_in:
            # weights of the extra channels, pre-multiplied by the BN alpha: [Cout, extra]
            self.c1_extra = (blk.conv1.weight.detach()[:, sl[1]:, 0, 0].float() * self.c1.scale[:, None]).contiguous()
            self.ds_extra = (blk.downsample[0].weight.detach()[:, sl[1]:, 0, 0].float() * self.ds.scale[:, None]).contiguous()


class ResNetEngine:
    """ResNet trunk (reference: networks/backbones.py:124-136)."""

    def __init__(self, prm, arena):
        self.arena = arena
        self.stem = conv_params(prm.conv1, prm.bn1, relu=True, stem4=True)
        self.stages = []
        for name in ("layer1", "layer2", "layer3"):
            self.stages.append([_BlockPlan(b, dtype=arena.dtype) for b in getattr(prm, name)])

    def _block(self, x, bp, tag, c1_shift=None, ds_shift=None):
        a = self.arena
        n, h, w, _ = x.shape
        ho = ops.conv_out_size(h, 1, bp.c1.stride, 0, 1)
        wo = ops.conv_out_size(w, 1, bp.c1.stride, 0, 1)
        if bp.ds is not None and c1_shift is None and ds_shift is None and 0 < n * ho * wo <= GROUP_MAX_ROWS and x.dtype == torch.float32:
            # small step: conv1 and the downsample conv read the same x -- one grouped launch
            y1, res = ops.conv2d_group([x, x], [bp.c1, bp.ds], [a.get("y1", (n, ho, wo, bp.c1.cout)),
                                                                a.get("res", (n, ho, wo, bp.ds.cout))])
            y2 = ops.conv2d(y1, bp.c2, out=a.get("y2", (n, ho, wo, bp.c2.cout)))
            return ops.conv2d(y2, bp.c3, out=a.get(("blk", tag), (n, ho, wo, bp.c3.cout)), residual=res)
        y1 = ops.conv2d(x, bp.c1, out=a.get("y1", (n, ho, wo, bp.c1.cout)),
                        shift_override=c1_shift, per_image_shift=c1_shift is not None)
        y2 = ops.conv2d(y1, bp.c2, out=a.get("y2", (n, ho, wo, bp.c2.cout)))
        if bp.ds is not None:
            res = ops.conv2d(x, bp.ds, out=a.get("res", (n, ho, wo, bp.ds.cout)),
                             shift_override=ds_shift, per_image_shift=ds_shift is not None)
        else:
            res = x
        return ops.conv2d(y2, bp.c3, out=a.get(("blk", tag), (n, ho, wo, bp.c3.cout)), residual=res)

    def stem_forward(self, x4):
        a = self.arena
        n, h, w, _ = x4.shape
        ho, wo = ops.conv_out_size(h, 7, 2, 3, 1), ops.conv_out_size(w, 7, 2, 3, 1)
        y = ops.conv2d(x4, self.stem, out=a.get("stem", (n, ho, wo, 64), torch.float32))
        hp, wp = ops._pool_out(ho, 3, 2, 1, True), ops._pool_out(wo, 3, 2, 1, True)
        x = ops.maxpool2d(y, 3, 2, 1, ceil_mode=True, out=a.get("pool", (n, hp, wp, 64), torch.float32))
        if a.dtype != torch.float32:       # bf16 variant: the fp32 stem + pool hand over a bf16 copy
            x = ops.convert(x, a.get("pool16", (n, hp, wp, 64)))
        return x

    def forward(self, x4):
        x = self.stem_forward(x4)
        for si, blocks in enumerate(self.stages):
            for bi, bp in enumerate(blocks):
                x = self._block(x, bp, (si, bi & 1))
        return x


class ResNetCMEngine(ResNetEngine):
    """ResNetCM (reference: networks/backbones.py:208-247).

    The communication module appends two channels that are CONSTANT over space (and over the
    S+Q images of an episode) before each stage.  A 1x1 conv over a constant channel is a
    per-image bias, so instead of concatenating, the first block of each stage receives
    ``shift[img][co] = beta[co] + alpha[co] * sum_e W[co][C+e] * feat[img][e]``.
    """

    def __init__(self, prm, arena):
        self.arena = arena
        self.spq = prm.spq
        self.stem = conv_params(prm.conv1, prm.bn1, relu=True, stem4=True)
        self.stages = []
        for name in ("layer1", "layer2", "layer3"):
            blocks = list(getattr(prm, name))
            self.stages.append([_BlockPlan(blocks[0], extra_in=2)] + [_BlockPlan(b) for b in blocks[1:]])
        self.lin = [(l.weight.detach().float().contiguous(), l.bias.detach().float().contiguous())
                    for l in (prm.linear1, prm.linear2, prm.linear3)]
        self.group = None   # LongTensor [N]: episode id of every image (set by the caller)

    def _comm(self, x, mask, lin, stride):
        """-> (feat [episodes,2], pooled mask): statistics, episode mean and the 2C -> 2 Linear on HIP kernels."""
        mask, stat = ops.cm_reduce(x, mask, stride)
        _, feat = ops.cm_linear(stat, self.group, lin[0], lin[1], self.n_groups)
        return feat, mask

    def forward(self, x4, prior):
        """x4: NHWC4 input (RGB + prior); prior: [N,H,W] fp32 mask plane."""
        mask = ops.cm_reduce(None, prior, 2)[0]                       # backbones.py:227
        x = self.stem_forward(x4)
        strides = (2, 1, 2)                                           # backbones.py:230,235,240
        for si, blocks in enumerate(self.stages):
            feat, mask = self._comm(x, mask, self.lin[si], strides[si])
            b0 = blocks[0]
            c1_shift = ops.cm_bias(feat, self.group, b0.c1_extra, base=b0.c1.shift)
            ds_shift = ops.cm_bias(feat, self.group, b0.ds_extra, base=b0.ds.shift)
            x = self._block(x, b0, (si, 0), c1_shift, ds_shift)
            for bi, bp in enumerate(blocks[1:], start=1):
                x = self._block(x, bp, (si, bi & 1))
        return x


class VGG16CMEngine:
    """VGG16CM (reference: networks/backbones.py:424-500).

    Unlike ResNetCM, whose communication channels enter 1x1 convs (a per-image bias), here they enter zero-padded 3x3
    convs: at the image border some taps of the constant planes read padding, so their contribution is not constant
    over space.  They are therefore really concatenated: every stage's pooled output is written into the first C
    channels of a [N,h,w,C+32] buffer (the max-pool kernel takes an output stride), the two broadcast values into
    channels C, C+1, and the next stage's first conv runs over C+32 input channels with zero weights on the 30 padding
    channels (the conv engine wants Cin % 32 == 0): 50 / 25 / 12 / 6 % more K on four of the thirteen convs."""
    PADC = 32

    def __init__(self, prm, arena):
        from .networks.backbones import VGG_CM_LAYOUT
        self.arena, self.spq = arena, prm.spq
        self.stages = []
        first = True
        for si, (name, nconv, cout, d, pool) in enumerate(VGG_CM_LAYOUT):
            seq = getattr(prm, name)
            convs = []
            for k in range(nconv):
                conv = seq[2 * k]
                relu = not (pool is None and k == nconv - 1 and not prm.last_relu)
                if k == 0 and not first:
                    convs.append(self._pack_extra(conv, relu))
                else:
                    convs.append(conv_params(conv, None, relu=relu, stem4=first and k == 0))
            self.stages.append((convs, cout, pool))
            first = False
        self.lin = [(l.weight.detach().float().contiguous(), l.bias.detach().float().contiguous())
                    for l in (prm.linear1, prm.linear2, prm.linear3, prm.linear4)]
        self.group, self.n_groups = None, 0

    def _pack_extra(self, conv, relu):
        """[Cout, C+2, 3, 3] -> KRSC over C + PADC input channels (zero weights on the padding channels)."""
        w = conv.weight.detach().float()
        co, ci, kh, kw = w.shape
        wp = torch.zeros((co, kh, kw, ci - 2 + self.PADC), dtype=torch.float32, device=w.device)
        wp[..., :ci] = w.permute(0, 2, 3, 1)
        cin = ci - 2 + self.PADC
        return ConvParams(wp.reshape(co, kh * kw * cin).contiguous(), None, conv.bias.detach().float().contiguous(), cin, co,
                          kh, kw, 1, conv.padding[0], conv.dilation[0], kh * kw * cin, False, relu)

    def forward(self, x4, prior):
        """x4: NHWC4 input (RGB + prior); prior: [N,H,W] fp32 mask plane -> NHWC features [N,h,w,512]."""
        a = self.arena
        x, mask = x4, prior
        for si, (convs, cout, pool) in enumerate(self.stages):
            for k, cp in enumerate(convs):
                n, h, w, _ = x.shape
                x = ops.conv2d(x, cp, out=a.get(("vcm", si, k & 1), (n, h, w, cp.cout)))
            if pool is None:
                return x
            n, h, w, c = x.shape
            ho, wo = ops._pool_out(h, 3, pool, 1, False), ops._pool_out(w, 3, pool, 1, False)
            wide = a.get(("vcm_cat", si), (n, ho, wo, c + self.PADC), zero=True)      # padding channels stay zero
            xs = ops.maxpool2d(x, 3, pool, 1, out=wide[..., :c])
            mask, stat = ops.cm_reduce(xs, mask, pool)                                # comm: mask pooled with the stage's stride
            _, feat = ops.cm_linear(stat, self.group, self.lin[si][0], self.lin[si][1], self.n_groups)
            wide[..., c:c + 2] = feat.index_select(0, self.group.long()).view(n, 1, 1, 2)   # broadcast over space (and the episode)
            x = wide
        return x


class ASPPV2Engine:
    """purifier.6 of stage 1 (reference: networks/backbones.py:359-369)."""

    def __init__(self, prm, arena):
        self.arena = arena
        self.bn = [bn_affine(getattr(prm, f"aspp_{i}")[0]) for i in range(5)]
        self.br = [conv_params(getattr(prm, f"aspp_{i}")[2], None, relu=True) for i in range(5)]
        midc = self.br[0].cout
        l6 = prm.layer6
        # layer6 split: columns of the broadcast global branch -> per-image bias; the rest -> 1x1 conv
        self.l6_global = conv_params(l6, None, relu=False, in_slice=(0, midc))
        self.l6_main = conv_params(l6, None, relu=False, in_slice=(midc, 5 * midc))
        self.l6_main.shift = None
        self.midc = midc
        # The BatchNorm in front of every branch folds into its conv: W*s, bias + sum_taps W t, and -- because the
        # reference zero-pads the BN OUTPUT -- out-of-image taps of the dilated convs read -t/s (ops.fold_input_affine).
        # That removes the pass that wrote four normalised copies of x (20 % of the non-conv time of an eval step).
        # PEMP_ASPP_COPIES=1 (or a zero BN scale) keeps the copy path.
        folded = [ops.fold_input_affine(self.br[i], *self.bn[i]) for i in range(5)]
        self.folded = None if (os.environ.get("PEMP_ASPP_COPIES") or any(f is None for f in folded)) else folded
        self._tails = set()
        if arena.dtype != torch.float32:
            # bf16 variant: the four spatial branches and layer6's main part take bf16 operands (folded in fp32 first, then
            # rounded once); the global branch works on [n, 1, 1, c] and stays fp32
            if self.folded is None:
                raise ValueError("the bf16 variant needs the folded ASPPV2 branches (non-zero BatchNorm scales)")
            for q, _ in self.folded[1:]:
                q.w = q.w.to(arena.dtype)
            self.l6_main.w = self.l6_main.w.to(arena.dtype)

    def forward(self, x, tail=None):
        """``tail`` [>= 4, c]: spare rows right behind ``x`` in the same allocation; the padding vectors are parked there
        (once per buffer) so that the buffer-addressed conv kernels reach them through the activations' descriptor."""
        a = self.arena
        n, h, w, c = x.shape
        midc = self.midc
        f32 = torch.float32
        x32 = x if x.dtype == f32 else ops.convert(x, a.get("aspp_x32", (n, h, w, c), f32))     # bf16 variant: the pooling reads fp32
        g = ops.global_avgpool(x32, out=a.get("gap", (n, c), f32))
        cat = a.get("aspp_cat", (n, h, w, 4 * midc))
        if self.folded is not None:
            g2 = ops.conv2d(g.view(n, 1, 1, c), self.folded[0][0], out=a.get("gap_c", (n, 1, 1, midc), f32))
            bias6 = ops.conv2d(g2, self.l6_global, out=a.get("bias6", (n, 1, 1, self.l6_global.cout), f32))
            if tail is not None and tail.data_ptr() not in self._tails:
                for i in range(4):
                    tail[i].copy_(self.folded[i + 1][1])
                self._tails.add(tail.data_ptr())
            qs = [self.folded[i + 1][0] for i in range(4)]
            pvs = [tail[i] if tail is not None else self.folded[i + 1][1] for i in range(4)]
            outs = [cat[..., i * midc:(i + 1) * midc] for i in range(4)]
            if tail is not None and 0 < n * h * w <= GROUP_MAX_ROWS and x.dtype == f32:
                # small step: the four branches read the same x -- one grouped launch, the dilated 3x3 convs first (their
                # tiles are the long ones; the 1x1 branch's short tiles fill in behind them)
                order = sorted(range(4), key=lambda i: -qs[i].kh * qs[i].kw)
                ops.conv2d_group([x] * 4, [qs[i] for i in order], [outs[i] for i in order], pad_values=[pvs[i] for i in order])
            else:
                for i in range(4):
                    ops.conv2d(x, qs[i], out=outs[i], pad_value=pvs[i] if qs[i].kh * qs[i].kw > 1 else None)
            return ops.conv2d(cat, self.l6_main, out=a.get("feat", (n, h, w, self.l6_main.cout), f32),
                              shift_override=bias6.view(n, -1), per_image_shift=True)
        gb = a.get("gap_bn", (n, c))
        ops.channel_affine_multi(g, [self.bn[0][0]], [self.bn[0][1]], [gb])
        g2 = ops.conv2d(gb.view(n, 1, 1, c), self.br[0], out=a.get("gap_c", (n, 1, 1, midc)))
        bias6 = ops.conv2d(g2, self.l6_global, out=a.get("bias6", (n, 1, 1, self.l6_global.cout)))
        xb = [a.get(("aspp_in", i), (n, h, w, c)) for i in range(4)]
        ops.channel_affine_multi(x, [s for s, _ in self.bn[1:]], [t for _, t in self.bn[1:]], xb)
        for i in range(4):
            ops.conv2d(xb[i], self.br[i + 1], out=cat[..., i * midc:(i + 1) * midc])
        return ops.conv2d(cat, self.l6_main, out=a.get("feat", (n, h, w, self.l6_main.cout)),
                          shift_override=bias6.view(n, -1), per_image_shift=True)


class ASPPEngine:
    """purifier.6 of stage 2: conv -> ReLU per branch, no BN (reference: backbones.py:310-321)."""

    def __init__(self, prm, arena):
        self.arena = arena
        self.br = [conv_params(getattr(prm, f"aspp_{i}")[0], None, relu=True) for i in range(5)]
        midc = self.br[0].cout
        self.l6_global = conv_params(prm.layer6, None, relu=False, in_slice=(0, midc))
        self.l6_main = conv_params(prm.layer6, None, relu=False, in_slice=(midc, 5 * midc))
        self.l6_main.shift = None
        self.midc = midc

    def forward(self, x):
        a = self.arena
        n, h, w, c = x.shape
        midc = self.midc
        g = ops.global_avgpool(x, out=a.get("gap", (n, c)))
        g2 = ops.conv2d(g.view(n, 1, 1, c), self.br[0], out=a.get("gap_c", (n, 1, 1, midc)))
        bias6 = ops.conv2d(g2, self.l6_global, out=a.get("bias6", (n, 1, 1, self.l6_global.cout)))
        cat = a.get("aspp_cat", (n, h, w, 4 * midc))
        outs = [cat[..., i * midc:(i + 1) * midc] for i in range(4)]
        if 0 < n * h * w <= GROUP_MAX_ROWS:          # small step: one grouped launch (see ASPPV2Engine.forward)
            order = sorted(range(4), key=lambda i: -self.br[i + 1].kh * self.br[i + 1].kw)
            ops.conv2d_group([x] * 4, [self.br[i + 1] for i in order], [outs[i] for i in order])
        else:
            for i in range(4):
                ops.conv2d(x, self.br[i + 1], out=outs[i])
        return ops.conv2d(cat, self.l6_main, out=a.get("feat", (n, h, w, self.l6_main.cout)),
                          shift_override=bias6.view(n, -1), per_image_shift=True)


class PurifierEngine:
    def __init__(self, seq, arena):
        self.arena = arena
        self.p0 = conv_params(seq[0], None, relu=True, dtype=arena.dtype)
        self.p3 = conv_params(seq[3], None, relu=True, dtype=arena.dtype)
        aspp = seq[6]
        self.aspp = ASPPV2Engine(aspp, arena) if isinstance(aspp.aspp_0[0], nn.BatchNorm2d) else ASPPEngine(aspp, arena)

    def forward(self, x):
        a = self.arena
        n, h, w, _ = x.shape
        y = ops.conv2d(x, self.p0, out=a.get("pur0", (n, h, w, self.p0.cout)))
        # the ASPP input carries 8 spare pixel rows behind it: the folded-BatchNorm branches keep their per-channel
        # padding vectors there, inside the buffer-descriptor window of the tensor (conv_dma2.hip, PADV)
        c = self.p3.cout
        flat = a.get("pur3+tail", (n * h * w + 8, c))
        y = ops.conv2d(y, self.p3, out=flat[:n * h * w].view(n, h, w, c))
        if isinstance(self.aspp, ASPPV2Engine):
            return self.aspp.forward(y, tail=flat[n * h * w:])
        return self.aspp.forward(y)


class VGG16Engine:
    """VGG16 trunk (reference: networks/backbones.py:404-405)."""

    def __init__(self, prm, arena):
        from .networks.backbones import VGG_LAYOUT
        self.arena = arena
        self.steps = []
        for item in VGG_LAYOUT:
            if isinstance(item, tuple):
                idx, _, _, _, relu = item
                conv = prm.features[idx]
                relu = relu or prm.last_relu
                self.steps.append(conv_params(conv, None, relu=relu, stem4=(idx == 0)))
            else:
                self.steps.append(item)

    def forward(self, x4):
        a = self.arena
        x = x4
        for i, st in enumerate(self.steps):
            n, h, w, _ = x.shape
            if isinstance(st, ConvParams):
                x = ops.conv2d(x, st, out=a.get(("vgg", i & 1, st.cout), (n, h, w, st.cout)))
            else:
                ho, wo = ops._pool_out(h, 3, st, 1, False), ops._pool_out(w, 3, st, 1, False)
                x = ops.maxpool2d(x, 3, st, 1, out=a.get(("vggp", i), (n, ho, wo, x.shape[3])))
        return x
