"""Training step of PEMP stage 1 (ResNet-50) on MI355X: explicit forward/backward over the HIP
kernels, flat parameter/gradient buffers, fused clip-norm + SGD, RCCL gradient all-reduce.

Counterpart of ``Trainer.train_step`` (reference entry/pemp_stage1.py:57-65) with the model in
``train()`` mode (core/base_trainer.py:189): BatchNorm normalises with BATCH statistics and updates
its running statistics (momentum 0.1), DropBlock is active (networks/pemp_stage1.py:76,79;
backbones.py:329-353), gradients are clipped to norm 1.1 and SGD(momentum 0.9, wd 5e-4) steps
(core/solver.py:87-91).

Data layout: every trainable parameter lives in ONE flat fp32 buffer (conv weights in KRSC order,
exposed to ``state_dict`` as channels_last views of the reference's [Cout,Cin,KH,KW] shape), every
gradient in a second flat buffer of the same layout -- that buffer is the single all-reduce bucket
(47.8 MB for stage 1) and the operand of the fused optimizer kernel.

The encoder (99.9 % of the flops) runs entirely on libpemp_hip.so: raw conv -> batch statistics ->
normalise(+residual)(+ReLU) forward; BN backward -> MFMA wgrad -> dgrad (the forward kernel with
flipped/transposed weights) backward.  The prototype head (MPM / cosine / upsample / CE; 0.1 % of
the flops) runs forward and backward on the HIP head kernels (head.hip, head_bwd.hip); the torch-autograd
restatement of the head that cross-checks them lives with the tests (tests/util.py).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops, train_ops as T
from .ops import ConvParams


def conv2d(x, p, **kw):
    """ops.conv2d for the training step: the autotuner may also pick the split-K variants (not bit-identical to the
    others, which the evaluation path relies on)."""
    return ops.conv2d(x, p, splitk=True, **kw)


FUSE_DROPBLOCK = os.environ.get("PEMP_FUSE_DROPBLOCK", "1") != "0"    # DropBlock's scaling inside the neighbouring kernel (A/B switch)

BN_MOM = 0.1
FUSE_BN_STATS = os.environ.get("PEMP_FUSE_BN_STATS", "1") != "0"   # conv epilogue starts the batch statistics (A/B switch)
FUSE_BN_BWD = os.environ.get("PEMP_FUSE_BN_BWD", "1") != "0"       # input-gradient epilogue starts the BatchNorm backward
SEG_EVERY = int(os.environ.get("PEMP_SEG_EVERY", "1"))     # residual blocks per graph segment of the backward pass


# ---------------------------------------------------------------------------------------------
# flat parameter storage
# ---------------------------------------------------------------------------------------------
def flat_layout(model):
    """-> (trainable parameters in ``model.parameters()`` order, their offsets in the flat buffers [floats, 16-byte aligned],
    total length).  Pure: no device memory (tests derive the all-reduce schedule of the real model from it on the CPU)."""
    params = [p for p in model.parameters() if p.requires_grad]
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    return params, offs, n


def stage1_bucket_cuts(model, params=None, offs=None, first_convs=None):
    """Candidate bucket boundaries of a model's flat gradient buffer: the offset of every residual block's first conv weight
    (forward order; ``first_convs``: those weights, default: ``model.encoder.backbone.layer1..3[*].conv1.weight``) + the offset
    where the purifier / ASPP parameters start (``GradBuckets`` forms the buckets from them) -> (block offsets, tail offset)."""
    if params is None:
        params, offs, _ = flat_layout(model)
    at = {id(p): o for p, o in zip(params, offs)}
    enc = getattr(model, "encoder", None)
    if first_convs is None:
        bb = getattr(enc, "backbone", None)
        first_convs = [blk.conv1.weight for name in ("layer1", "layer2", "layer3") for blk in getattr(bb, name, [])]
    block_off = [at.get(id(w), 0) for w in first_convs]
    pur = getattr(enc, "purifier", None)
    tail_off = min([at[id(p)] for p in pur.parameters() if id(p) in at], default=0) if pur is not None else 0
    return block_off, tail_off


class FlatParams:
    """Re-homes the trainable parameters of ``model`` into one flat buffer (+ one for gradients)."""

    def __init__(self, model, device):
        self.params, offs, n = flat_layout(model)
        self.n = n
        self.data = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        self._mom = None               # SGD momentum, allocated by the first SGD step (an Adam run never needs it)
        self.offs = offs
        self.first_step = True
        for p, o in zip(self.params, offs):
            src = p.detach().to(device)
            self._view(self.data, p, o).copy_(src)
            p.data = self._view(self.data, p, o)
        self.grad_views = [self._view(self.grad, p, o) for p, o in zip(self.params, offs)]
        self.attach_grads()
        # weight-gradient kernels run on a side stream (see _Conv.wgrad); own workspace cache, joined by the engine
        self.side_stream = (ops.concurrent_stream(torch.device(device), priority=int(os.environ.get("PEMP_SIDE_PRIORITY", "0")))
                            if torch.device(device).type == "cuda" else None)      # verified to run BESIDE the current stream
        self.side_ws, self.side_keep = {}, []
        self.capture = None            # a SegmentedCapture while the step is being recorded (Stage1Trainer.use_graph)

    @property
    def mom(self):
        if self._mom is None:
            self._mom = torch.zeros_like(self.data)
        return self._mom

    def cut(self):
        """Segment point of the backward pass (no-op unless a segmented capture is recording): everything enqueued so far
        becomes one main-stream graph, the weight-gradient launches deferred since the last cut one side-stream graph."""
        if self.capture is not None:
            self.capture.cut()

    def join_side_stream(self):
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)
        self.side_keep.clear()

    @staticmethod
    def _view(buf, p, o):
        """The slice of flat buffer ``buf`` that holds parameter ``p``, shaped like ``p`` (conv weights: KRSC storage
        seen through a channels_last view of the reference's [Cout,Cin,KH,KW] shape)."""
        if p.dim() == 4:
            co, ci, kh, kw = p.shape
            return buf[o:o + p.numel()].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return buf[o:o + p.numel()].view(p.shape)

    def attach_grads(self):
        """(Re)attach every ``p.grad`` as a view of the flat gradient buffer (``zero_grad(set_to_none=True)`` drops them)."""
        for p, v in zip(self.params, self.grad_views):
            p.grad = v

    def gather_grads(self):
        """Flat copy of the gradients currently held in ``p.grad`` (None when there are none): lets a backward pass
        ADD to them, as autograd does."""
        if all(p.grad is None for p in self.params):
            return None
        if all(p.grad is v for p, v in zip(self.params, self.grad_views)):
            return self.grad.clone()
        prev = torch.zeros_like(self.grad)
        for p, o in zip(self.params, self.offs):
            if p.grad is not None:
                self._view(prev, p, o).copy_(p.grad)
        return prev

    def build_dgrad_mirror(self):
        """A second flat buffer holding, for every conv weight, the KRSC weight of its input-gradient conv ([Cin, KH*KW*Cout],
        taps flipped) at the same offset; refreshed by ONE kernel launch per step (pemp_dgrad_mirror_f32: tiled transposes
        of all layers) instead of a transpose per layer."""
        self.dg_view, layers = {}, []
        for p, o in zip(self.params, self.offs):
            if p.dim() != 4:
                continue
            co, ci, kh, kw = p.shape
            layers.append((o, co, kh * kw, ci))
            self.dg_view[id(p)] = (o, ci, kh * kw * co)
        self.dg_table, self.dg_tiles = T.dgrad_mirror_table(layers, self.data.device)
        self.dg_data = torch.zeros_like(self.data)

    def refresh_dgrad_mirror(self):
        """Runs on the side stream: only the backward pass reads the mirror (``backward`` joins first), so the launch
        hides under the forward pass."""
        if self.side_stream is None or self.capture is not None:
            T.dgrad_mirror(self.data, self.dg_data, self.dg_table, self.dg_tiles)
            return
        self.side_stream.wait_stream(torch.cuda.current_stream())     # after the optimizer step / the last dgrad reader
        with torch.cuda.stream(self.side_stream):
            T.dgrad_mirror(self.data, self.dg_data, self.dg_table, self.dg_tiles)

    def dgrad_krsc(self, p):
        o, rows, cols = self.dg_view[id(p)]
        return self.dg_data[o:o + rows * cols].view(rows, cols)

    def krsc(self, p):
        """[Cout, KH*KW*Cin] KRSC matrix view of a conv weight (no copy)."""
        co, ci, kh, kw = p.shape
        return p.data.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)

    def krsc_grad(self, p):
        co, ci, kh, kw = p.shape
        return p.grad.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)


# ---------------------------------------------------------------------------------------------
# layer records
# ---------------------------------------------------------------------------------------------
class SegmentedCapture:
    """The training step as a CHAIN of hipGraphs that keeps the two-stream overlap.

    One hipGraph of the whole step replays its two branches one after the other (measured: 23.2 ms against 21.5 ms eager),
    and the eager step costs ~16 ms of host enqueue time for ~450 launches.  Here the step is cut at ``FlatParams.cut()``
    points of the backward pass into segments; segment k yields a main-stream graph M_k (forward / input-gradient /
    BatchNorm chain) and a side-stream graph S_k holding the weight-gradient launches deferred during M_k.  Replay:

        main:  M_0 -> e_0 -> M_1 -> e_1 -> M_2 ... -> M_n -> wait(side)
        side:        wait(e_0) -> S_0 -> wait(e_1) -> S_1 ...              (S_k runs under M_{k+1})

    i.e. ~2n graph launches per step instead of ~450 kernel launches, with the same concurrency as the eager step at
    segment granularity.  All graphs share one memory pool; every tensor a deferred weight-gradient reads is kept alive
    until the last segment has been captured (its memory must not be handed to a later segment that runs beside it)."""

    def __init__(self, flat):
        self.flat = flat
        self.main, self.side, self.keep, self.pending = [], [], [], []
        self.cs = torch.cuda.Stream(device=flat.data.device)           # capture stream of the main chain
        # two memory pools: graphs that share a pool must never run beside each other (a block freed during one capture
        # is handed out again in the next), and S_k runs beside M_{k+1}
        self.pool, self.side_pool = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        self.cur = None
        self.events = None

    def begin(self):
        torch.cuda.synchronize()
        self.cs.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self.cs)
        self._ctx.__enter__()
        self.flat.capture = self
        self._start()

    def _start(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool)

    def defer(self, conv, x, g):
        self.pending.append((conv, x, g))

    def cut(self, last=False):
        self.cur.capture_end()
        self.main.append(self.cur)
        sg = None
        if self.pending:
            sg = torch.cuda.CUDAGraph()
            side = self.flat.side_stream
            side.wait_stream(self.cs)
            with torch.cuda.stream(side):
                sg.capture_begin(pool=self.side_pool)
                for conv, x, g in self.pending:
                    conv._wgrad_now(x, g, self.flat.side_ws)
                sg.capture_end()
            self.cs.wait_stream(side)
            self.keep += self.pending
            self.pending = []
        self.side.append(sg)
        if not last:
            self._start()

    def end(self):
        self.cut(last=True)
        self.flat.capture = None
        self._ctx.__exit__(None, None, None)
        torch.cuda.current_stream().wait_stream(self.cs)
        torch.cuda.synchronize()
        self.keep = []                     # all segments recorded: nothing can be handed their memory any more
        self.events = [torch.cuda.Event() for _ in self.main]

    def abort(self):
        """The recorded step raised: leave capture mode with the trainer usable again (the caller re-raises the original
        error).  The open graph is ended -- an invalidated capture raises here, which is swallowed: the first error is the
        one that matters -- nothing recorded so far is kept, weight gradients stop being deferred, the capture stream's
        context is left."""
        try:
            if self.cur is not None:
                self.cur.capture_end()
        except Exception:  # noqa: BLE001
            pass
        self.cur = None
        self.main, self.side, self.keep, self.pending = [], [], [], []
        self.flat.capture = None
        self._ctx.__exit__(None, None, None)
        torch.cuda.current_stream().wait_stream(self.cs)

    def replay(self):
        cur, side = torch.cuda.current_stream(), self.flat.side_stream
        side.wait_stream(cur)              # the side graphs must not start before this step's inputs / zeroed gradients
        for m, sg, ev in zip(self.main, self.side, self.events):
            m.replay()
            if sg is not None:
                ev.record(cur)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    sg.replay()
        cur.wait_stream(side)


def _enqueue_wgrad(flat, job, x, g, ws):
    """Run ``job._wgrad_now(x, g, workspace cache)`` where weight gradients run: deferred into the side graph of the current
    segment while a SegmentedCapture is recording, on the side stream behind an event otherwise, in place without one."""
    if flat.capture is not None:
        flat.capture.defer(job, x, g)
        return
    if flat.side_stream is None or x.shape[0] * x.shape[1] * x.shape[2] < 64:
        return job._wgrad_now(x, g, ws)
    ready = torch.cuda.Event()
    ready.record()
    flat.side_keep.append((x, g))                      # operands stay referenced until the join
    with torch.cuda.stream(flat.side_stream):
        flat.side_stream.wait_event(ready)
        job._wgrad_now(x, g, flat.side_ws)


class _SliceWgrad:
    """Weight gradient of a 1x1 conv whose weight is a column slice of a larger matrix (ASPPV2's layer6 sees the global branch
    and the four concatenated branches as two such slices): computed into a dense temporary, copied into the slice."""

    def __init__(self, dst, cin, cout):
        self.dst, self.prm = dst, ConvParams(None, None, None, cin, cout, 1, 1, 1, 0, 1, cin, False, False)

    def _wgrad_now(self, x, g, ws):
        tmp = torch.empty((self.prm.cout, self.prm.cin), dtype=torch.float32, device=x.device)
        T.conv_wgrad(x, g, self.prm, tmp, ws_cache=ws)
        self.dst.copy_(tmp)


class _Conv:
    """Geometry + parameter handles of one conv; packed views are refreshed every step."""

    def __init__(self, flat, conv, stem=False):
        self.flat, self.conv, self.stem = flat, conv, stem
        w = conv.weight
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.stride, self.pad, self.dil = conv.stride[0], conv.padding[0], conv.dilation[0]
        self.trainable = w.requires_grad

    def fwd_params(self, relu=False, with_bias=True):
        w = self.conv.weight
        if self.stem:
            packed, kpad = ops.pack_conv_weight(w.data, stem4=True)
            cin = 4
        else:
            packed = self.flat.krsc(w) if self.trainable else w.data.permute(0, 2, 3, 1).reshape(self.cout, -1).contiguous()
            kpad, cin = packed.shape[1], self.cin
        bias = self.conv.bias.data if (with_bias and self.conv.bias is not None) else None
        return ConvParams(packed, None, bias, cin, self.cout, self.kh, self.kw, self.stride, self.pad, self.dil, kpad,
                          self.stem, relu)

    def dgrad_params(self):
        wd = self.flat.dgrad_krsc(self.conv.weight)          # refreshed once per step by the engine
        return ConvParams(wd, None, None, self.cout, self.cin, self.kh, self.kw, 1, self.dil * (self.kh - 1) - self.pad,
                          self.dil, wd.shape[1], False, False)

    def wgrad(self, x, g, ws):
        """Writes conv.weight.grad (and nothing else).  Nothing in the backward pass depends on a weight gradient,
        so it is enqueued on the engine's side stream (after the event that says ``g`` is ready) and runs concurrently
        with the input-gradient / BatchNorm chain, filling the CUs those small-M kernels leave idle; the engine joins
        the side stream at the end of ``backward``."""
        _enqueue_wgrad(self.flat, self, x, g, ws)

    def _wgrad_now(self, x, g, ws):
        w = self.conv.weight
        if self.stem:
            dw = torch.empty((self.cout, 256), dtype=torch.float32, device=x.device)
            prm = ConvParams(None, None, None, 4, self.cout, self.kh, self.kw, self.stride, self.pad, self.dil, 256, True, False)
            T.conv_wgrad(x, g, prm, dw, ws_cache=ws)
            k = self.kh * self.kw
            w.grad.copy_(dw[:, :k * 4].view(self.cout, self.kh, self.kw, 4)[..., :self.cin].permute(0, 3, 1, 2))
            return
        prm = ConvParams(None, None, None, self.cin, self.cout, self.kh, self.kw, self.stride, self.pad, self.dil,
                         self.kh * self.kw * self.cin, False, False)
        T.conv_wgrad(x, g, prm, self.flat.krsc_grad(w), ws_cache=ws)


class _BN:
    def __init__(self, bn):
        self.bn = bn

    def stats(self, z, ws):
        return T.bn_stats(z, self.bn.eps, BN_MOM, self.bn.running_mean, self.bn.running_var, ws_cache=ws)

    def stats_from(self, part, m):
        return T.bn_stats_partials(part, m, self.bn.eps, BN_MOM, self.bn.running_mean, self.bn.running_var)

    def grad_out(self):
        """(dgamma, dbeta) destinations inside the flat gradient buffer, or None for a frozen BatchNorm."""
        return (self.bn.weight.grad, self.bn.bias.grad) if self.bn.weight.requires_grad else None

    def write_grads(self, dgamma, dbeta):
        if self.bn.weight.requires_grad:
            self.bn.weight.grad.copy_(dgamma)
            self.bn.bias.grad.copy_(dbeta)


class Stage1TrainEngine:
    """Forward + backward of the stage-1 ResNet-50 encoder in train mode on HIP kernels."""

    def __init__(self, model, device):
        self.model, self.device = model, device
        self.flat = FlatParams(model, device)
        for b in model.buffers():
            b.data = b.data.to(device)
        for p in model.parameters():
            if not p.requires_grad:
                p.data = p.data.to(device)
        self.ws = {}
        self.drop_rate, self.block_size = 0.0, 4
        self.draws = None                                            # see _dropblock
        self.rng = T.RandomStream(torch.initial_seed(), device)      # Philox stream of the DropBlock / Dropout2d kernels
        self._init_trunk(model.encoder.backbone)
        self._init_tail(model)
        self.flat.build_dgrad_mirror()
        self.bn_counters = [m.num_batches_tracked for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
        # overlapped gradient all-reduce: bucket boundaries at residual-block starts (see GradBuckets)
        # (the purifier / ASPP parameters are laid out after the backbone and finished first)
        self.block_off, self.tail_off = stage1_bucket_cuts(model, self.flat.params, self.flat.offs,
                                                           first_convs=[b["c1"].conv.weight for b in getattr(self, "blocks", [])])
        self.buckets = GradBuckets(self.flat.grad, self.block_off + [self.tail_off], side_stream=self.flat.side_stream)

    def _init_trunk(self, bb):
        f = self.flat
        self.stem = (_Conv(f, bb.conv1, stem=True), _BN(bb.bn1))
        self.blocks = []
        for name in ("layer1", "layer2", "layer3"):
            for blk in getattr(bb, name):
                self.blocks.append(dict(
                    c1=_Conv(f, blk.conv1), b1=_BN(blk.bn1), c2=_Conv(f, blk.conv2), b2=_BN(blk.bn2),
                    c3=_Conv(f, blk.conv3), b3=_BN(blk.bn3),
                    ds=(_Conv(f, blk.downsample[0]), _BN(blk.downsample[1])) if blk.downsample is not None else None))

    def _init_tail(self, model):
        f, pur = self.flat, model.encoder.purifier
        self.p0, self.p3 = _Conv(f, pur[0]), _Conv(f, pur[3])
        aspp = pur[6]
        self.aspp_bn = [_BN(getattr(aspp, f"aspp_{i}")[0]) for i in range(5)]
        self.aspp_conv = [_Conv(f, getattr(aspp, f"aspp_{i}")[2]) for i in range(5)]
        self.l6 = aspp.layer6
        self.midc = self.aspp_conv[0].cout

    # -- helpers ------------------------------------------------------------------------------
    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _dropblock(self, x, n, h, w, layer):
        """DropBlock2D(drop_rate, block_size) in train(): -> (y, record for the backward) (identity at rate 0).  ``layer`` is
        the module's name in the reference model; ``self.draws`` ({layer: uniforms [n,h,w]}) replaces the Philox stream by
        given draws (parity tests: the same draws go to the oracle's restatement of the layer)."""
        rec = self._dropblock_rec(n, h, w, layer)
        return (x, None) if rec is None else (T.pixel_scale(x, *rec), rec)

    def _dropblock_rec(self, n, h, w, layer):
        """The mask + kept count of one DropBlock2D call (None at rate 0): what ``_dropblock`` applies, and what the fused
        forms (conv epilogue, BatchNorm apply) take instead of a pass of their own."""
        if self.drop_rate <= 0.0:
            return None
        u = None
        if self.draws is not None:
            u = self.draws[layer]
            if tuple(u.shape) != (n, h, w) or u.dtype != torch.float32 or u.device != self.flat.data.device or not u.is_contiguous():
                raise ValueError(f"dropblock draws of {layer}: want contiguous float32 {(n, h, w)} on {self.flat.data.device}")
        return T.dropblock_mask(n, h, w, self.drop_rate, self.block_size, self.rng, self.device, uniforms=u)

    @staticmethod
    def _dropblock_bwd(dy, rec):
        return dy if rec is None else T.pixel_scale(dy, *rec)

    def _cbn_fwd(self, x, conv, bn, relu, residual=None, img_bias=None):
        """conv -> (+ per-image bias [N,Cout]) -> batch-stat BN (+residual)(+ReLU); returns (y, tape record)."""
        prm = conv.fwd_params(relu=False, with_bias=False)
        part = None
        if FUSE_BN_STATS and img_bias is None and ops.stats_supported(x, prm):
            z, part = ops.conv2d_stats(x, prm)         # batch statistics started in the conv epilogue
        else:
            z = conv2d(x, prm) if img_bias is None else conv2d(x, prm, shift_override=img_bias, per_image_shift=True)
        mask = None
        if FUSE_BN_BWD and relu and T.mask_supported(z.shape[-1]):       # sign bits of y: what the backward needs of it
            mask = torch.empty((z.numel() // z.shape[-1], z.shape[-1] // 32), dtype=torch.int32, device=z.device)
        if part is not None:                           # ... finished and applied by one call
            y, mean, invstd = T.bn_fwd_partials(z, part, bn.bn.weight.data, bn.bn.bias.data, torch.empty_like(z), bn.bn.eps, BN_MOM,
                                                bn.bn.running_mean, bn.bn.running_var, residual=residual, relu=relu, mask=mask)
            return y, dict(x=x, z=z, y=y, mean=mean, invstd=invstd, relu=relu, mask=mask)
        mean, invstd = bn.stats(z, self.ws)
        y = T.bn_apply(z, mean, invstd, bn.bn.weight.data, bn.bn.bias.data, torch.empty_like(z), residual=residual, relu=relu,
                       mask=mask)
        return y, dict(x=x, z=z, y=y, mean=mean, invstd=invstd, relu=relu, mask=mask)

    def _cbn_bwd(self, dy, rec, conv, bn, want_gout=False, need_dx=True, add_to=None, up=None):
        """-> (dx [compact for stride-2], gout).  ``add_to`` is added to dx inside the dgrad epilogue; the
        gradient at the conv output stays in rec["dz"].  ``up``: the tape record of the BatchNorm whose OUTPUT gradient dx
        is (the conv -> bn -> relu in front of this one): the dgrad epilogue then masks dx with that BatchNorm's ReLU and
        starts its column sums (ops.conv2d_bnbwd), and up["gpart"] tells its _cbn_bwd to finish from there."""
        dz = torch.empty_like(rec["z"])
        if "gpart" in rec:           # dy is the masked gradient g already; its sums were started by the conv that made it
            T.bn_bwd_partials(dy, rec["z"], rec["mean"], rec["invstd"], bn.bn.weight.data, rec.pop("gpart"), dz,
                              out=bn.grad_out())
            gout = dy if want_gout else None
        else:
            gout = torch.empty_like(rec["z"]) if want_gout else None
            T.bn_bwd(dy, rec["y"], rec["z"], rec["mean"], rec["invstd"], bn.bn.weight.data, dz, gout=gout, relu=rec["relu"],
                     ws_cache=self.ws, out=bn.grad_out(),        # dgamma / dbeta go straight into the flat gradient buffer
                     mask=rec.get("mask"))
        conv.wgrad(rec["x"], dz, self.ws)
        rec["dz"] = dz
        if not need_dx:
            return None, gout
        prm = conv.dgrad_params()
        if (up is not None and FUSE_BN_BWD and (up["mask"] is not None or not up["relu"]) and prm.shift is None
                and ops.stats_supported(dz, prm)):
            dx, up["gpart"] = ops.conv2d_bnbwd(dz, prm, up, residual=add_to)
        else:
            dx = conv2d(dz, prm, residual=add_to)
        return dx, gout

    def _block_fwd(self, x, b, bias_c1=None, bias_ds=None):
        rec = {"x": x}
        y1, rec["r1"] = self._cbn_fwd(x, b["c1"], b["b1"], True, img_bias=bias_c1)
        y2, rec["r2"] = self._cbn_fwd(y1, b["c2"], b["b2"], True)
        if b["ds"] is not None:
            res, rec["rd"] = self._cbn_fwd(x, b["ds"][0], b["ds"][1], False, img_bias=bias_ds)
        else:
            res = x
        out, rec["r3"] = self._cbn_fwd(y2, b["c3"], b["b3"], True, residual=res)
        return out, rec

    def _block_bwd(self, dx, b, rec, up=None):
        """``up``: tape record of the BatchNorm(+ReLU) that produced this block's input (the previous block's bn3)."""
        x = rec["x"]
        stride = b["c1"].stride
        dy2, gout = self._cbn_bwd(dx, rec["r3"], b["c3"], b["b3"], want_gout=True, up=rec["r2"])
        dy1, _ = self._cbn_bwd(dy2, rec["r2"], b["c2"], b["b2"], up=rec["r1"])
        if stride != 1:
            up = None                # the compact stride-2 gradient is scattered first: not the producer's final value
        if b["ds"] is None:
            return self._cbn_bwd(dy1, rec["r1"], b["c1"], b["b1"], add_to=gout, up=up)[0]
        dxd, _ = self._cbn_bwd(gout, rec["rd"], b["ds"][0], b["ds"][1])
        dxc, _ = self._cbn_bwd(dy1, rec["r1"], b["c1"], b["b1"], add_to=dxd, up=up)
        return dxc if stride == 1 else T.scatter_strided(dxc, (x.shape[1], x.shape[2]), stride)

    # -- forward ------------------------------------------------------------------------------
    def forward(self, images_list, **kw):
        """images_list: [n_i,3,H,W] tensors -> NHWC features; keeps what backward needs in self.tape."""
        tape = {}
        if self.bn_counters:
            torch._foreach_add_(self.bn_counters, 1)        # every BatchNorm runs exactly once per step
        self.flat.refresh_dgrad_mirror()
        self.rng.begin_step()          # a device-side increment: also advances inside a replayed hipGraph
        x = self._trunk_forward(images_list, tape, **kw)
        feat = self._tail_forward(x, tape)
        self.tape = tape
        return feat

    def backward(self, dfeat):
        recording = self.flat.capture is not None
        if self.flat.side_stream is not None and not recording:      # the side stream must not start before this step's gradients were zeroed
            torch.cuda.current_stream().wait_stream(self.flat.side_stream)     # the dgrad weight mirror is in place
            self.flat.side_stream.wait_stream(torch.cuda.current_stream())
        dx = self._tail_backward(dfeat, up=self.tape["blocks"][-1]["r3"] if self.tape.get("blocks") else None)
        self.flat.cut()                            # segment 0: forward + head + purifier / ASPP backward
        if self.tail_off:
            self.buckets.ready_from(self.tail_off)       # purifier / ASPP gradients are final: all-reduce under layer3's backward
        self._trunk_backward(dx)
        if not recording:
            self.flat.join_side_stream()           # every weight gradient has landed before the optimizer / all-reduce
        self.tape = None

    def _pack(self, images_list, priors=None):
        n = sum(t.shape[0] for t in images_list)
        H, W = images_list[0].shape[-2:]
        x4 = self._new(n, H, W, 4)
        o = 0
        for i, t in enumerate(images_list):
            ops.pack_input(t.contiguous(), None if priors is None else priors[i].contiguous(), out=x4[o:o + t.shape[0]])
            o += t.shape[0]
        return x4

    def _trunk_forward(self, images_list, tape):
        y, tape["stem"] = self._cbn_fwd(self._pack(images_list), *self.stem, relu=True)
        x, tape["pool_idx"] = T.maxpool_idx(y, 3, 2, 1, ceil_mode=True)
        tape["pool_in"], tape["blocks"] = y, []
        for b in self.blocks:
            x, rec = self._block_fwd(x, b)
            tape["blocks"].append(rec)
        return x

    def _tail_forward(self, x, tape):
        # purifier: conv+bias+ReLU (+DropBlock) twice
        nimg, h, w, _ = x.shape
        if FUSE_DROPBLOCK:
            # conv -> bias -> ReLU -> DropBlock in ONE launch each (the layer's scaling in the conv's epilogue).  The backward pass
            # needs ReLU's sign of the conv output only where the DropBlock kept the pixel, and there sign(xa) = sign(ya) (the
            # scale is positive): xa / xb stand in for ya / yb on the tape.
            da = self._dropblock_rec(nimg, h, w, "encoder.purifier.2")
            ya = xa = conv2d(x, self.p0.fwd_params(relu=True), dropblock=da)
            db = self._dropblock_rec(nimg, h, w, "encoder.purifier.5")
            yb = xb = conv2d(xa, self.p3.fwd_params(relu=True), dropblock=db)
        else:
            ya = conv2d(x, self.p0.fwd_params(relu=True))
            xa, da = self._dropblock(ya, nimg, h, w, "encoder.purifier.2")
            yb = conv2d(xa, self.p3.fwd_params(relu=True))
            xb, db = self._dropblock(yb, nimg, h, w, "encoder.purifier.5")
        tape.update(p0_in=x, ya=ya, da=da, xa=xa, yb=yb, db=db, xb=xb)
        # ASPPV2: five BNs share the statistics of xb (branch 0: of its global average)
        midc = self.midc
        gap = ops.global_avgpool(xb)
        m0, i0 = self.aspp_bn[0].stats(gap, self.ws)
        t0 = T.bn_apply(gap, m0, i0, self.aspp_bn[0].bn.weight.data, self.aspp_bn[0].bn.bias.data, torch.empty_like(gap), relu=False)
        t0d, d0 = self._dropblock(t0, nimg, 1, 1, "encoder.purifier.6.aspp_0.1")
        g0 = conv2d(t0d.view(nimg, 1, 1, -1), self.aspp_conv[0].fwd_params(relu=True))
        l6w = self.l6.weight
        w6 = self.flat.krsc(l6w)                                   # [512, 1280]
        w6g = ConvParams(w6[:, :midc].contiguous(), None, self.l6.bias.data, midc, l6w.shape[0], 1, 1, 1, 0, 1, midc, False, False)
        bias6 = conv2d(g0, w6g)
        cat = self._new(nimg, h, w, 4 * midc)
        ts, ds_ = [], []
        mean_x = invstd_x = None
        for i in range(1, 5):
            bn = self.aspp_bn[i]
            # the four BNs see the same input, hence the same batch statistics; each call also moves
            # that BN's own running statistics
            mean_x, invstd_x = bn.stats(xb, self.ws)
            d = self._dropblock_rec(nimg, h, w, f"encoder.purifier.6.aspp_{i}.1") if FUSE_DROPBLOCK else None
            if d is not None and T.mask_supported(xb.shape[-1]):           # BatchNorm apply + DropBlock in one pass
                td = T.bn_apply_dropblock(xb, mean_x, invstd_x, bn.bn.weight.data, bn.bn.bias.data, torch.empty_like(xb), d)
            else:
                t = T.bn_apply(xb, mean_x, invstd_x, bn.bn.weight.data, bn.bn.bias.data, torch.empty_like(xb), relu=False)
                td, d = (T.pixel_scale(t, *d), d) if d is not None else self._dropblock(t, nimg, h, w, f"encoder.purifier.6.aspp_{i}.1")
            conv2d(td, self.aspp_conv[i].fwd_params(relu=True), out=cat[..., (i - 1) * midc:i * midc])
            ts.append(td)
            ds_.append(d)
        w6m = ConvParams(w6[:, midc:].contiguous(), None, None, 4 * midc, l6w.shape[0], 1, 1, 1, 0, 1, 4 * midc, False, False)
        feat = conv2d(cat, w6m, shift_override=bias6.view(nimg, -1), per_image_shift=True)
        tape.update(gap=gap, m0=m0, i0=i0, t0d=t0d, d0=d0, g0=g0, cat=cat, ts=ts, ds=ds_, mean_x=mean_x, invstd_x=invstd_x,
                    w6=w6, hw=(nimg, h, w))
        return feat

    # -- backward -----------------------------------------------------------------------------
    def _tail_backward(self, dfeat, up=None):
        """``up``: tape record of the last residual block's bn3 (the BatchNorm + ReLU whose output the tail consumes): the
        purifier's input-gradient conv then starts that BatchNorm's backward in its epilogue (see _cbn_bwd)."""
        tp, midc = self.tape, self.midc
        nimg, h, w = tp["hw"]
        hw = h * w
        l6w = self.l6.weight
        w6 = tp["w6"]
        dw6 = self.flat.krsc_grad(l6w)                              # [512, 1280] view of the gradient
        # layer6: main 1x1 conv over the 4 concatenated branches + per-image bias from the global branch
        _enqueue_wgrad(self.flat, _SliceWgrad(dw6[:, midc:], 4 * midc, l6w.shape[0]), tp["cat"], dfeat, self.ws)   # side stream
        dcat = conv2d(dfeat, ConvParams(T.dgrad_weight(w6[:, midc:].contiguous(), 1, 1), None, None, l6w.shape[0], 4 * midc,
                                            1, 1, 1, 0, 1, l6w.shape[0], False, False))
        s = ops.global_avgpool(dfeat) * float(hw)                   # per-image column sums [N, 512]
        self.l6.bias.grad.copy_(s.sum(dim=0))
        _enqueue_wgrad(self.flat, _SliceWgrad(dw6[:, :midc], midc, l6w.shape[0]), tp["g0"], s.view(nimg, 1, 1, -1), self.ws)
        dg0 = conv2d(s.view(nimg, 1, 1, -1), ConvParams(T.dgrad_weight(w6[:, :midc].contiguous(), 1, 1), None, None, l6w.shape[0],
                                                            midc, 1, 1, 1, 0, 1, l6w.shape[0], False, False))
        # branches 1..4
        dxb = None
        for i in range(1, 5):
            conv, bn = self.aspp_conv[i], self.aspp_bn[i]
            u = tp["cat"][..., (i - 1) * midc:i * midc]
            g = self._new(nimg, h, w, midc)
            T.relu_bias_bwd(dcat[..., (i - 1) * midc:i * midc], u, g, relu=True, ws_cache=self.ws, out=conv.conv.bias.grad)
            conv.wgrad(tp["ts"][i - 1], g, self.ws)
            if FUSE_DROPBLOCK and tp["ds"][i - 1] is not None:          # DropBlock's backward in the input-gradient conv's epilogue
                dt = conv2d(g, conv.dgrad_params(), dropblock=tp["ds"][i - 1])
            else:
                dt = self._dropblock_bwd(conv2d(g, conv.dgrad_params()), tp["ds"][i - 1])
            dz = self._new(nimg, h, w, dt.shape[-1])
            T.bn_bwd(dt, None, tp["xb"], tp["mean_x"], tp["invstd_x"], bn.bn.weight.data, dz, relu=False, ws_cache=self.ws,
                     out=bn.grad_out())                  # dgamma / dbeta straight into the flat gradient buffer
            if dxb is None:
                dxb = dz
            else:
                T.relu_bias_bwd(dz, None, dxb, add=dxb, relu=False, want_dbias=False)
        # branch 0 (global)
        conv0, bn0 = self.aspp_conv[0], self.aspp_bn[0]
        g = self._new(nimg, 1, 1, midc)
        T.relu_bias_bwd(dg0, tp["g0"], g, relu=True, ws_cache=self.ws, out=conv0.conv.bias.grad)
        conv0.wgrad(tp["t0d"].view(nimg, 1, 1, -1), g, self.ws)
        dt0 = conv2d(g, conv0.dgrad_params()).view(nimg, -1)
        dt0 = self._dropblock_bwd(dt0, tp["d0"])
        dgap = self._new(nimg, dt0.shape[1])
        T.bn_bwd(dt0, None, tp["gap"], tp["m0"], tp["i0"], bn0.bn.weight.data, dgap, relu=False, ws_cache=self.ws, out=bn0.grad_out())
        T.gap_bwd_add(dgap, dxb)
        # purifier.3 and purifier.0 (conv + bias + ReLU (+ DropBlock))
        dxb = self._dropblock_bwd(dxb, tp["db"])
        g = torch.empty_like(tp["yb"])
        T.relu_bias_bwd(dxb, tp["yb"], g, relu=True, ws_cache=self.ws, out=self.p3.conv.bias.grad)
        self.p3.wgrad(tp["xa"], g, self.ws)
        if FUSE_DROPBLOCK and tp["da"] is not None:
            dxa = conv2d(g, self.p3.dgrad_params(), dropblock=tp["da"])
        else:
            dxa = self._dropblock_bwd(conv2d(g, self.p3.dgrad_params()), tp["da"])
        g = torch.empty_like(tp["ya"])
        T.relu_bias_bwd(dxa, tp["ya"], g, relu=True, ws_cache=self.ws, out=self.p0.conv.bias.grad)
        self.p0.wgrad(tp["p0_in"], g, self.ws)
        prm = self.p0.dgrad_params()
        if up is not None and FUSE_BN_BWD and up["mask"] is not None and prm.shift is None and ops.stats_supported(g, prm):
            dx, up["gpart"] = ops.conv2d_bnbwd(g, prm, up)
            return dx
        return conv2d(g, prm)

    def _trunk_backward(self, dx):
        tp = self.tape
        for bi in range(len(self.blocks) - 1, -1, -1):                          # residual blocks, last to first
            dx = self._block_bwd(dx, self.blocks[bi], tp["blocks"][bi], up=tp["blocks"][bi - 1]["r3"] if bi > 0 else None)
            self.buckets.ready_from(self.block_off[bi])
            if bi % SEG_EVERY == SEG_EVERY - 1:
                self.flat.cut()                    # every SEG_EVERY-th block: its weight gradients run under the next blocks' chain
        # stem: max pool, BN+ReLU, 7x7 conv (weight gradient only)
        dy = T.maxpool_idx_bwd(tp["pool_idx"], dx, tp["pool_in"].shape[1:3], 3, 2, 1)
        self._cbn_bwd(dy, tp["stem"], *self.stem, need_dx=False)


class GradBuckets:
    """Bucketed SUM all-reduce of the flat gradient buffer, overlapped with the backward pass.

    The backward pass finishes gradients from the END of the flat buffer towards its start (parameters are laid out
    in forward order: ctr, stem, layer1..3, purifier/ASPP).  ``cuts`` are candidate bucket boundaries (block starts);
    buckets of >= ``min_bytes`` are formed from the end.  The engine calls ``ready_from(lo)`` whenever every gradient at
    offsets >= lo is final (its kernels are enqueued); each bucket that lies wholly above ``lo`` is then all-reduced
    asynchronously -- on GPU under the side stream, so that RCCL's stream waits for the weight-gradient kernels while
    the main stream goes on with the input-gradient chain.  ``finish()`` launches what is left, waits for everything
    and returns the factor that turns the sum into the mean.  Every rank launches the same buckets in the same order.
    xGMI is point-to-point: a ring all-reduce of B bytes over 8 GPUs moves 1.75 B per GPU, so the 47.8 MB of
    stage 1 cost ~0.5 ms unoverlapped; in 3-6 buckets all but the last (the 5.8 MB of ctr..layer2) hide under layer3..1."""

    def __init__(self, flat_grad, cuts, min_bytes=8 << 20, side_stream=None):
        self.grad, self.side = flat_grad, side_stream
        n = flat_grad.numel()
        bounds, hi = [n], n
        for c in sorted({int(c) for c in cuts if 0 < int(c) < n}, reverse=True):
            if (hi - c) * 4 >= min_bytes and c * 16 >= min_bytes:       # a full bucket above, no sliver (< 1/4 bucket) below
                bounds.append(c)
                hi = c
        bounds.append(0)
        self.buckets = [(bounds[i + 1], bounds[i]) for i in range(len(bounds) - 1) if bounds[i + 1] < bounds[i]]   # descending
        self.next, self.pending = 0, []
        self.enabled = False          # the Trainer that also calls finish() switches the hooks on

    def active(self):
        """Hooks fire only for a trainer that will call finish(), in a process group of > 1 ranks
        (PEMP_FORCE_BUCKETS=1: also with one rank, to exercise the stream choreography on a single GPU)."""
        return (self.enabled and dist.is_available() and dist.is_initialized()
                and (dist.get_world_size() > 1 or bool(os.environ.get("PEMP_FORCE_BUCKETS"))))

    def _launch(self, lo, hi):
        chunk = self.grad[lo:hi]
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record()                                   # gradients written by the main stream (BN, bias, ctr)
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):            # ... and by the weight-gradient kernels queued on the side stream
                self.pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))

    def ready_from(self, lo):
        if not self.active() or (self.side is not None and torch.cuda.is_current_stream_capturing()):
            return
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= lo:
            self._launch(*self.buckets[self.next])
            self.next += 1

    def finish(self):
        if not self.active():
            self.next = 0
            return 1.0
        self.ready_from(0)
        for w in self.pending:
            w.wait()                                      # the current stream waits for the collective
        self.next, self.pending = 0, []
        return 1.0 / dist.get_world_size()


def allreduce_gradients(flat_grad):
    """The one collective of a training step: SUM all-reduce of the flat gradient bucket (47.8 MB for
    stage 1) over RCCL (gloo on CPU in tests).  Returns the factor that turns the sum into the mean;
    the fused optimizer kernel applies it, so no extra pass over the bucket is needed."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()
    return 1.0


class Stage1Trainer:
    """``train_step`` with the reference's contract: zero_grad, forward, CE loss, backward,
    clip_grad_norm_(1.1), SGD step; returns the loss tensor (entry/pemp_stage1.py:57-65)."""

    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=5e-4, max_norm=1.1, device=None,
                 drop_rate=None, block_size=None, loss="ce", sigma=5.0, use_graph=False):
        from .networks.pemp_stage1 import net_ingredient
        cfg = net_ingredient.cfg
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.model = model
        model.train()
        if getattr(model, "backbone_name", "resnet50") == "vgg16":     # stage 1 on VGG-16: conv+bias+ReLU chain, no purifier
            from .train_baseline import _VGGEngine
            self.eng = _VGGEngine(model, self.device)
        else:
            self.eng = Stage1TrainEngine(model, self.device)
        self.eng.drop_rate = getattr(model, "drop_rate", cfg["drop_rate"]) if drop_rate is None else drop_rate
        self.eng.block_size = getattr(model, "block_size", cfg["block_size"]) if block_size is None else block_size
        self.lr, self.momentum, self.wd, self.max_norm = lr, momentum, weight_decay, max_norm
        self.protos = 0 if model.ctr is None else model.ctr.shape[1] // 2
        self.dist_scalar = cfg["dist_scalar"]
        self.last_grad_norm = None
        self.nesterov = False
        self.optimizer = None                     # optional torch optimizer acting as hyper-parameter / scheduler holder
        self.use_graph = use_graph
        self._graphs = {}
        from .core import losses
        self.loss_obj = losses.get({"loss": loss, "sigma": sigma})

    map_full_res = False          # Baseline: masked average pooling over bilinearly up-sampled features
    #: False = a rank-LOCAL step: no gradient collective at all (bucket hooks off, no all-reduce in the optimizer step).
    #: For passes that only one rank runs (bench.py's instrumented roofline pass): a collective issued by one rank alone
    #: would pair with whatever the other ranks issue next.
    collectives = True
    #: a list: ``reduce_gradients`` appends one timing record per step (bench.py's ``comm.exposed_comm_ms``); None: not timed
    comm_log = None

    def encode(self, sup_img, sup_mask, qry_img, qry_prior=None):
        """Train-mode encoder forward -> NHWC features [B*S + B*Q, h, w, c] (supports first); tape kept in the engine."""
        B, S, ch, H, W = sup_img.shape
        Q = qry_img.shape[1]
        return self.eng.forward([sup_img.reshape(B * S, ch, H, W), qry_img.reshape(B * Q, ch, H, W)])

    def forward_backward(self, sup_img, sup_mask, qry_img, qry_msk):
        """Fills the flat gradient buffer; returns (loss, low-resolution prediction): encoder, prototype head and
        their backward all on libpemp_hip.so (the logits are never materialised: CE and its gradient are fused
        with the bilinear upsample)."""
        B, S = sup_img.shape[:2]
        Q = qry_img.shape[1]
        self.eng.flat.attach_grads()
        self.eng.flat.grad.zero_()
        feat = self.encode(sup_img, sup_mask, qry_img)
        return self._head_hip(feat, sup_mask, qry_msk, B, S, Q)

    def _head_hip(self, feat, sup_mask, qry_msk, B, S, Q):
        """MPM / cosine / upsample + CE forward and backward on the HIP kernels."""
        if Q != 1:
            raise ValueError("query must be 1")
        eng, ws = self.eng, self.eng.ws
        H, W = sup_mask.shape[-2:]
        msk = sup_mask.reshape(B * S, 2, H, W).contiguous()
        tgt = qry_msk.reshape(-1, *qry_msk.shape[-2:]).contiguous()
        sup, qry = feat[:B * S], feat[B * S:]
        ctr = self.model.ctr
        if self.protos > 0:
            key = ("mpm", B, S, sup.shape[1], sup.shape[2], sup.shape[3], self.protos)
            pro = ops.mpm_protos(sup, msk, ctr.data, B, S, self.protos, ws_cache=ws)
        else:
            key = ("map", B, S, sup.shape[1], sup.shape[2], sup.shape[3])
            pro = ops.masked_avg_pool(sup, msk, B, S, full_res=False, ws_cache=ws)
        pred = ops.cosine_proto_max(qry, pro, self.dist_scalar)
        wmap = self.loss_obj.weight_map(tgt)                    # None for plain CE
        _, stats, _ = ops.eval_tail(pred, tgt, ws_cache=ws, weight=wmap)
        loss = stats[:, 0].sum() / stats[:, 1].sum()
        dfeat = torch.empty_like(feat)
        dctr = T.head_bwd(sup, qry, msk, ctr.data if ctr is not None else None, ws[key], pro, pred, tgt, stats, dfeat,
                          B, S, self.protos, self.dist_scalar, ws_cache=ws, weight=wmap)
        if ctr is not None:
            ctr.grad.copy_(dctr)
        eng.backward(dfeat)
        return loss.float(), pred

    def train_step(self, sup_img, sup_mask, qry_img, qry_msk=None):
        ins = (sup_img.to(self.device), sup_mask.to(self.device), qry_img.to(self.device), qry_msk.to(self.device))
        self.eng.buckets.enabled = self.collectives and not self.use_graph   # eager step: buckets are all-reduced during backward
        loss = self._graphed_forward_backward(*ins) if self.use_graph else self.forward_backward(*ins)[0]
        self.optimizer_step()                               # finishes / waits for the buckets
        self.eng.buckets.enabled = False
        return loss

    def _graphed_forward_backward(self, *ins):
        """forward + backward (~450 short launches) replayed from a chain of hipGraphs per input signature
        (``SegmentedCapture``: main-stream segments + side-stream weight-gradient segments, overlap preserved); the first
        two calls run eagerly (they populate workspaces and the conv autotune cache).  The optimizer (all-reduce + fused
        clip/SGD) stays outside the graphs."""
        key = tuple((tuple(t.shape), t.dtype) for t in ins)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {"calls": 0}
        if "cap" not in ent:
            ent["calls"] += 1
            if ent["calls"] <= 2:
                return self.forward_backward(*ins)[0]
            static = [t.clone() for t in ins]
            cap = SegmentedCapture(self.eng.flat)
            cap.begin()
            try:
                loss, _ = self.forward_backward(*static)
                cap.end()
            except BaseException:
                cap.abort()                # keeps the original error; two eager calls later the step is recorded again
                ent["calls"] = 0
                raise
            ent.update(cap=cap, static=static, loss=loss)
        for s_, t in zip(ent["static"], ins):
            s_.copy_(t, non_blocking=True)
        ent["cap"].replay()
        return ent["loss"].clone()          # the graphs' output buffer is overwritten by the next replay

    def attach_optimizer(self, optimizer):
        """Use ``optimizer.param_groups[0]`` (lr, momentum, weight_decay, nesterov) -- e.g. the object returned by
        ``pemp_amd.core.solver.get`` with its LR scheduler -- as the source of hyper-parameters of every step."""
        self.optimizer = optimizer

    def reduce_gradients(self):
        """The step's gradient collective: finishes / waits for the buckets launched during backward (or all-reduces the whole
        flat buffer when none was) and returns the factor that turns the SUM into the mean; 1.0 without touching the
        process group for a rank-local step (``collectives`` False)."""
        if not self.collectives:
            return 1.0
        log = self.comm_log
        if log is None:
            return self.eng.buckets.finish() if self.eng.buckets.next else allreduce_gradients(self.eng.flat.grad)
        # measured: how long the step's own stream is held by the collective(s) it has to wait for here -- the EXPOSED part of
        # the gradient exchange (the buckets launched during the backward pass ran under it)
        if self.device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            scale = self.eng.buckets.finish() if self.eng.buckets.next else allreduce_gradients(self.eng.flat.grad)
            e1.record()
            log.append((e0, e1))
        else:
            import time
            t0 = time.perf_counter()
            scale = self.eng.buckets.finish() if self.eng.buckets.next else allreduce_gradients(self.eng.flat.grad)
            log.append((time.perf_counter() - t0) * 1e3)
        return scale

    def exposed_comm_ms(self):
        """Mean over the logged steps (``comm_log``) of the time ``reduce_gradients`` held the step's stream; None without a log."""
        log = self.comm_log
        if not log:
            return None
        if isinstance(log[0], tuple):
            log[-1][1].synchronize()
            return sum(a.elapsed_time(b) for a, b in log) / len(log)
        return sum(log) / len(log)

    def optimizer_step(self):
        if self.optimizer is not None:
            g = self.optimizer.param_groups[0]
            self.lr, self.momentum, self.wd = g["lr"], g.get("momentum", 0.0), g.get("weight_decay", 0.0)
            self.nesterov = bool(g.get("nesterov", False))
        self.apply_update(self.reduce_gradients())

    def optimizer_state(self):
        """What the fused optimizer kernels keep beside the flat parameters (``optimizer.state`` of an attached torch object
        stays empty): SGD momentum, or Adam's moments and step count -- flat tensors in the layout of ``flat_layout``."""
        f = self.eng.flat
        st = {"first_step": f.first_step, "momentum": None if f._mom is None else f._mom.clone()}
        if getattr(f, "exp_avg", None) is not None:
            st.update(exp_avg=f.exp_avg.clone(), exp_avg_sq=f.exp_avg_sq.clone(), adam_step=f.adam_step)
        return st

    def load_optimizer_state(self, st):
        f = self.eng.flat
        f.first_step = bool(st["first_step"])
        if st.get("momentum") is not None:
            f.mom.copy_(st["momentum"])
        if "exp_avg" in st:
            if getattr(f, "exp_avg", None) is None:
                f.exp_avg, f.exp_avg_sq = torch.zeros_like(f.data), torch.zeros_like(f.data)
            f.exp_avg.copy_(st["exp_avg"])
            f.exp_avg_sq.copy_(st["exp_avg_sq"])
            f.adam_step = int(st["adam_step"])

    def apply_update(self, scale):
        f = self.eng.flat
        if (type(self.optimizer) is torch.optim.Adam and len(self.optimizer.param_groups) == 1
                and not any(g.get("amsgrad") or g.get("maximize") for g in self.optimizer.param_groups)):
            # (an Adam with several groups -- per-layer lr / weight decay -- steps through torch below: the fused kernel has
            # one set of hyper-parameters for the whole flat buffer)
            # tr.opt = adam (core/solver.py:92-96): fused clip + Adam on the flat buffers; the torch object holds the
            # hyper-parameters (and its LR scheduler), the moments live beside the flat parameters
            g = self.optimizer.param_groups[0]
            if getattr(f, "exp_avg", None) is None:
                f.exp_avg, f.exp_avg_sq, f.adam_step = torch.zeros_like(f.data), torch.zeros_like(f.data), 0
            f.adam_step += 1
            self.last_grad_norm = T.adam_clip_step(f.data, f.grad, f.exp_avg, f.exp_avg_sq, f.adam_step, self.max_norm, g["lr"],
                                                   g["betas"], g["eps"], g.get("weight_decay", 0.0), grad_scale=scale,
                                                   ws_cache=self.eng.ws)
            return
        if self.optimizer is not None and not isinstance(self.optimizer, torch.optim.SGD):
            # any other torch optimizer steps on the parameter views itself
            if scale != 1.0:
                f.grad.mul_(scale)
            f.attach_grads()
            if self.max_norm > 0:
                self.last_grad_norm = torch.nn.utils.clip_grad_norm_(f.params, self.max_norm)
            self.optimizer.step()
            return
        self.last_grad_norm = T.sgd_clip_step(f.data, f.grad, f.mom, self.max_norm, self.lr, self.momentum, self.wd,
                                              f.first_step, grad_scale=scale, ws_cache=self.eng.ws, nesterov=self.nesterov)
        f.first_step = False
