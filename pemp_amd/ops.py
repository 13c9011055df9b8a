"""Tensor-level wrappers over the C ABI (include/pemp_hip.h).

PyTorch is plumbing here: it owns device memory and the current stream; every function passes
raw device pointers + the current HIP stream to libpemp_hip.so.  Activations are NHWC fp32
tensors ``[N,H,W,C]`` whose last-dim stride is 1; a channel slice of a wider buffer is passed
as a view (its pixel stride ``ld`` is taken from ``stride(2)``).
"""
import ctypes as C
import json
import os

import torch

from . import _lib
from ._lib import ConvDesc, CONV_RELU, CONV_SHIFT_PER_IMAGE, CONV_STEM4  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk_dev(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda):
            raise _lib.PempHipError("pemp_amd ops need device (cuda/HIP) tensors; there is no CPU path")


def _nhwc(t, name, dtype=torch.float32):
    if t.dim() != 4 or t.dtype != dtype or t.stride(3) != 1:
        raise ValueError(f"{name}: expected {dtype} NHWC view with unit channel stride, got {tuple(t.shape)} {t.dtype} {t.stride()}")
    n, h, w, c = t.shape
    # strides of size-1 dims are arbitrary in torch: take the pixel stride from the first dim that moves
    if w > 1:
        ld = t.stride(2)
    elif h > 1:
        ld = t.stride(1)
    elif n > 1:
        ld = t.stride(0)
    else:
        ld = c
    if (w > 1 and h > 1 and t.stride(1) != w * ld) or (h * w > 1 and n > 1 and t.stride(0) != h * w * ld):
        raise ValueError(f"{name}: pixels must be densely packed with stride ld={ld}, got strides {t.stride()}")
    return ld


def conv_out_size(i, k, s, p, d):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


class ConvParams:
    """Device-resident, pre-packed parameters of one conv (+ folded per-channel affine)."""
    __slots__ = ("w", "scale", "shift", "cin", "cout", "kh", "kw", "stride", "pad", "dil", "kpad", "stem", "relu")

    def __init__(self, w, scale, shift, cin, cout, kh, kw, stride, pad, dil, kpad, stem, relu):
        self.w, self.scale, self.shift = w, scale, shift
        self.cin, self.cout, self.kh, self.kw = cin, cout, kh, kw
        self.stride, self.pad, self.dil, self.kpad, self.stem, self.relu = stride, pad, dil, kpad, stem, relu


#: kernel variants the autotuner may pick: id -> (BM, BN); ids >= 11 stage through LDS-DMA.  All variants
#: accumulate in the same K order, so they are bit-identical and the choice only affects speed.
TILE_VARIANTS = {13: (64, 64), 14: (128, 128), 12: (128, 64), 11: (128, 128), 15: (128, 64), 3: (64, 64),
                 17: (256, 256), 16: (256, 128),    # 16/17: 8-wave blocks, 98/131 KB LDS, half the L2->LDS bytes per flop
                 # 2x: the same shapes on conv_dma2.hip (buffer-addressed LDS-DMA, barrier inside the MFMA stream)
                 23: (64, 64), 24: (128, 128), 22: (128, 64), 21: (128, 128), 25: (128, 64), 27: (256, 256), 26: (256, 128),
                 # 28: 32 x 64 blocks of four 16 x 32 wave tiles on v_mfma_f32_16x16x4_f32 -- same K order, bit-identical (the fp32
                 # MFMAs are sequential fma chains: scratch/mfma_eq); finer granularity for launches of a few rounds (one episode)
                 28: (32, 64),
                 # 29: hybrid launch for convs of a few rounds -- the rows that fill whole rounds of the chip on the 64 x 64 tile, the
                 # remaining rows on 16-row wave tiles, one grid (csrc/conv_dma2.hip); other geometries run as 23
                 29: (64, 64)}
AUTOTUNE = True
#: test hook: ``PICK_HOOK(kind, cands, key) -> one of cands`` decides every kernel-variant pick INSTEAD of timing (kind "conv":
#: tile ids, "wgrad": block counts / (tile kind, block count) pairs), at any problem size.  Timing-based picks differ from box
#: to box and the split-K / weight-gradient splits change the rounding: a whole-step parity test pins them (tests/conftest.py
#: ``pinned_picks``: AUTOTUNE off = one fixed variant per layer) or sweeps them through this hook.
PICK_HOOK = None
SPLITK = os.environ.get("PEMP_CONV_SPLITK", "1") != "0"     # the training convs may pick the split-K variants (A/B switch)
DEFAULT_TILE = 13
_TILE_CACHE = {}     # (layer geometry, input shape) -> fastest variant; shared by every ConvParams object
#: optional JSON file the picks are loaded from / saved to (PEMP_TILE_CACHE=path): a profiling run can then replay a
#: previous process' choices instead of timing the variants again under the profiler
_TILE_CACHE_FILE = os.environ.get("PEMP_TILE_CACHE")
WGRAD_PICKS = {}     # train_ops.conv_wgrad's per-shape picks ((tile kind, block count)); persisted in the same file
if _TILE_CACHE_FILE and os.path.exists(_TILE_CACHE_FILE):
    with open(_TILE_CACHE_FILE) as _f:
        for _k, _v in json.load(_f).items():
            _key = json.loads(_k)
            if _key and _key[0] == "wgrad":
                WGRAD_PICKS[tuple(_key[1:])] = tuple(_v) if isinstance(_v, list) else int(_v)
            else:
                _TILE_CACHE[tuple(_key)] = int(_v)


def save_picks():
    """Write the conv tile picks and the weight-gradient picks to PEMP_TILE_CACHE (no-op without it): a later process --
    a profiling run, or a training run that must reproduce this one bit for bit -- replays them instead of timing again."""
    if not _TILE_CACHE_FILE:
        return
    out = {json.dumps([int(v) for v in k]): t for k, t in _TILE_CACHE.items()}
    out.update({json.dumps(["wgrad"] + [int(v) for v in k]): (list(t) if isinstance(t, tuple) else t) for k, t in WGRAD_PICKS.items()})
    with open(_TILE_CACHE_FILE, "w") as f:
        json.dump(out, f)


def concurrent_stream(device, priority=0, tries=6, us=200):
    """A new HIP stream that really runs beside the CURRENT one.  The runtime deals streams round-robin to a few hardware queues
    (four by default): the n-th stream a process creates can land on the queue of the stream it is meant to overlap with, and
    the two then run one after the other (seen: the training step lost its two-stream overlap, 16.6 -> 17.8 ms, whenever
    three other streams had been created first).  Candidates are created until two idle kernels of ``us`` microseconds, one
    on each stream, finish in clearly less than 2 x ``us``; the last candidate is returned if none does."""
    lib = _lib.load()
    cur = torch.cuda.current_stream(device)
    _lib.check(lib.pemp_spin_us(1, C.c_void_p(cur.cuda_stream)), "pemp_spin_us")      # the kernel's first launch (code load) is not timed
    cur.synchronize()
    keep = []                                   # rejected candidates stay alive until the choice is made (no index reuse)
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device, priority=priority)
        keep.append(cand)
        cand.wait_stream(cur)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        _lib.check(lib.pemp_spin_us(us, C.c_void_p(cur.cuda_stream)), "pemp_spin_us")
        _lib.check(lib.pemp_spin_us(us, C.c_void_p(cand.cuda_stream)), "pemp_spin_us")
        cur.wait_stream(cand)
        e1.record(cur)
        e1.synchronize()
        if e0.elapsed_time(e1) * 1e3 < 1.6 * us:
            return cand
    return keep[-1]


def export_picks():
    """Everything the autotuners have decided so far, as one picklable object (see ``tuned_by_rank0``)."""
    return {"tiles": dict(_TILE_CACHE), "wgrad": dict(WGRAD_PICKS)}


def import_picks(picks):
    _TILE_CACHE.update(picks["tiles"])
    WGRAD_PICKS.update(picks["wgrad"])


def tuned_by_rank0(warm):
    """Run ``warm()`` -- one untimed step that meets every conv / weight-gradient shape of the job -- so that only RANK 0
    times kernel variants: it warms first, its picks are broadcast (one ``broadcast_object_list`` of a few KB), the other
    ranks then warm with every shape already in the cache.  All ranks run the same variants afterwards (the split-K and
    weight-gradient split choices change the rounding: data-parallel replicas should not differ in them), and N - 1 ranks do
    not spend their warm-up on timing runs.  One process / no process group: just ``warm()``."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        warm()
        return
    rank = dist.get_rank()
    if rank == 0:
        warm()
    box = [export_picks() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    if rank != 0:
        import_picks(box[0])
        warm()


def _tunes(rows, least=1024):
    """Whether a variant pick happens now: the autotuner on a problem worth timing, or the test hook (any size)."""
    return (PICK_HOOK is not None or (AUTOTUNE and rows >= least)) and not torch.cuda.is_current_stream_capturing()


def _pick_tile(launch, p, key, cout, only=None):
    """Time the candidate variants for this (layer, input shape) and remember the fastest: two rounds over all
    candidates (the minimum of a variant's two timings counts: a round can be disturbed by whatever else the GPU is
    finishing), then a run-off between the best three with more repetitions."""
    if only is None:
        cands = [t for t, (bm, bn) in TILE_VARIANTS.items() if cout % bn == 0]
    else:
        cands = [t for t in only if cout % TILE_VARIANTS[t - 10 if t > 30 else t][1] == 0]
    if PICK_HOOK is not None:
        best = PICK_HOOK("conv", list(cands), key)
        if best not in cands:
            raise ValueError(f"PICK_HOOK returned {best!r}, not one of {cands}")
        _TILE_CACHE[key] = best
        return best

    def timed(t, reps):
        launch(t)                                   # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch(t)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    ms = {t: timed(t, 3) for t in cands}
    for t in cands:
        ms[t] = min(ms[t], timed(t, 3))
    top = sorted(cands, key=lambda t: ms[t])[:3]
    final = {t: min(timed(t, 8), timed(t, 8)) for t in top}
    best = min(top, key=lambda t: final[t])
    _TILE_CACHE[key] = best
    save_picks()
    return best


def hybrid_rows(n, ho, wo, cout):
    """Rows of an [n, ho, wo, cout] conv output that the hybrid launch (tile id 29) gives to the 64 x 64 tile; 0 = no split."""
    d = ConvDesc(n, ho, wo, 32, 32, ho, wo, cout, cout, 1, 1, 1, 0, 1, 0, 32, 0, 29)
    return int(_lib.load().pemp_conv2d_hybrid_rows(C.byref(d)))


def pack_conv_weight(w_oihw, stem4=False):
    """[Cout,Cin,KH,KW] -> KRSC [Cout, Kpad] (Cin contiguous).  STEM4: Cin padded to 4, row padded x32."""
    co, ci, kh, kw = w_oihw.shape
    w = w_oihw.detach().permute(0, 2, 3, 1).contiguous().float()     # [co,kh,kw,ci]
    if stem4:
        if ci > 4:
            raise ValueError("stem4 packing needs Cin <= 4")
        k = kh * kw * 4
        kpad = (k + 31) // 32 * 32
        out = torch.zeros(co, kpad, dtype=torch.float32, device=w.device)
        tmp = torch.zeros(co, kh, kw, 4, dtype=torch.float32, device=w.device)
        tmp[..., :ci] = w
        out[:, :k] = tmp.reshape(co, k)
        return out, kpad
    return w.reshape(co, kh * kw * ci), kh * kw * ci


def conv2d(x, p, out=None, residual=None, shift_override=None, per_image_shift=False, relu=None, tile=0,
           pad_value=None, splitk=False, dropblock=None):
    """y = act(scale * conv(x, w) + shift (+ residual)).  x: NHWC view, returns NHWC tensor/view ``out``.
    ``pad_value`` [Cin]: what out-of-image taps read instead of zero (multi-tap convs; see fold_input_affine).
    ``splitk``: the autotuner may also pick the split-K variants (training path: they are not bit-identical to the rest).
    ``dropblock`` (mask fp32 [N,Ho,Wo], kept count int32 [1]) -- a DropBlock2D record of train_ops.dropblock_mask: the layer's
    scaling of the output rows happens in the conv's epilogue (pemp_conv2d_dropblock_nhwc_f32; same arithmetic as
    train_ops.pixel_scale on the conv's result)."""
    if x.dtype == torch.bfloat16:
        return _conv2d_bf16(x, p, out, residual, shift_override, per_image_shift, relu, tile, pad_value)
    lib = _lib.load()
    _chk_dev(x, p.w, out, residual)
    ldx = _nhwc(x, "x")
    n, h, w, cin = x.shape
    if cin != p.cin:
        raise ValueError(f"conv2d: input has {cin} channels, layer expects {p.cin}")
    ho = conv_out_size(h, p.kh, p.stride, p.pad, p.dil)
    wo = conv_out_size(w, p.kw, p.stride, p.pad, p.dil)
    if out is None:
        out = torch.empty((n, ho, wo, p.cout), dtype=torch.float32, device=x.device)
    ldy = _nhwc(out, "out")
    if tuple(out.shape) != (n, ho, wo, p.cout):
        raise ValueError(f"conv2d: out shape {tuple(out.shape)} != {(n, ho, wo, p.cout)}")
    ldr = 0
    if residual is not None:
        ldr = _nhwc(residual, "residual")
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError("conv2d: residual shape mismatch")
    shift = p.shift if shift_override is None else shift_override
    if pad_value is not None:
        _chk_dev(pad_value)
        if pad_value.numel() != cin or pad_value.dtype != torch.float32 or not pad_value.is_contiguous():
            raise ValueError(f"conv2d: pad_value must be a contiguous fp32 [{cin}] vector")
    flags = 0
    if (p.relu if relu is None else relu):
        flags |= CONV_RELU
    if per_image_shift:
        flags |= CONV_SHIFT_PER_IMAGE
    if p.stem:
        flags |= CONV_STEM4
    splitk = ((splitk and SPLITK) or (EVAL_SPLITK and n * ho * wo <= EVAL_SPLITK_MAX_ROWS)) and not p.stem

    if dropblock is not None:
        dmask, dcnt = dropblock
        _chk_dev(dmask, dcnt)
        if (pad_value is not None or p.stem or dmask.dtype != torch.float32 or not dmask.is_contiguous() or dmask.numel() != n * ho * wo
                or dcnt.dtype != torch.int32 or not dma2_supported(x, p)):
            # outside the buffer-addressed kernels: conv, then the layer's own pass
            from . import train_ops
            y = conv2d(x, p, out=out, residual=residual, shift_override=shift_override, per_image_shift=per_image_shift, relu=relu,
                       tile=tile, pad_value=pad_value, splitk=splitk)
            return train_ops.pixel_scale(y, dmask, dcnt, out=y)

    def launch(t):
        d = ConvDesc(n, h, w, cin, ldx, ho, wo, p.cout, ldy, p.kh, p.kw, p.stride, p.pad, p.dil, ldr, p.kpad, flags, t)
        if dropblock is not None:
            ws, ws_bytes = _splitk_ws(lib, d, x.device) if t > 30 else (None, 0)
            _check_sk(lib, lib.pemp_conv2d_dropblock_nhwc_f32(C.byref(d), _p(x), _p(p.w), _p(out), _p(p.scale), _p(shift), _p(residual),
                                                              _p(dropblock[0]), _p(dropblock[1]), C.c_void_p(ws), ws_bytes, _stream()),
                      ws, "pemp_conv2d_dropblock_nhwc_f32")
            return
        if t > 30 and pad_value is not None:
            ws, ws_bytes = _splitk_ws(lib, d, x.device)
            _check_sk(lib, lib.pemp_conv2d_padv_splitk_nhwc_f32(C.byref(d), _p(x), _p(p.w), _p(out), _p(p.scale), _p(shift), _p(residual),
                                                                _p(pad_value), C.c_void_p(ws), ws_bytes, _stream()), ws,
                      "pemp_conv2d_padv_splitk_nhwc_f32")
        elif t > 30:
            ws, ws_bytes = _splitk_ws(lib, d, x.device)
            _check_sk(lib, lib.pemp_conv2d_splitk_nhwc_f32(C.byref(d), _p(x), _p(p.w), _p(out), _p(p.scale), _p(shift), _p(residual),
                                                           C.c_void_p(ws), ws_bytes, _stream()), ws, "pemp_conv2d_splitk_nhwc_f32")
        elif pad_value is None:
            _lib.check(lib.pemp_conv2d_nhwc_f32(C.byref(d), _p(x), _p(p.w), _p(out), _p(p.scale), _p(shift),
                                                _p(residual), _stream()), "pemp_conv2d_nhwc_f32")
        else:
            _lib.check(lib.pemp_conv2d_padv_nhwc_f32(C.byref(d), _p(x), _p(p.w), _p(out), _p(p.scale), _p(shift),
                                                     _p(residual), _p(pad_value), _stream()), "pemp_conv2d_padv_nhwc_f32")

    if tile == 0:
        key = (p.cin, p.cout, p.kh, p.kw, p.stride, p.pad, p.dil, (6 if dropblock is not None else 4) if splitk else int(p.stem) + (7 if dropblock is not None else 0),
               n, h, w, int(residual is not None), int(pad_value is not None))
        tile = _TILE_CACHE.get(key)
        if tile is None:
            if dropblock is not None:
                only = list(GROUP_TILES) + (list(SPLITK_TILES) if splitk else [])
            else:
                only = list(TILE_VARIANTS) + list(SPLITK_TILES) if splitk else None
            if _tunes(n * ho * wo):
                if only is None:
                    only = list(TILE_VARIANTS)
                if 29 in only and not (hybrid_rows(n, ho, wo, p.cout) and dma2_supported(x, p)):
                    only = [t for t in only if t != 29]       # no hybrid launch for this geometry / layer: id 29 would run as 23 (or 13)
                tile = _pick_tile(launch, p, key, p.cout, only=only)
            else:
                tile = DEFAULT_TILE + (10 if dropblock is not None else 0)
    launch(tile)
    return out


def _conv2d_bf16(x, p, out, residual, shift_override, per_image_shift, relu, tile, pad_value):
    """The bf16-operand variant of ``conv2d`` (pemp_conv2d_bf16_nhwc; the side figure of bench.py, never the default path): x,
    p.w, residual and pad_value are bf16, accumulation is fp32, ``out`` is bf16 -- or fp32 when an fp32 ``out`` is given (the
    encoder's last layer)."""
    lib = _lib.load()
    _chk_dev(x, p.w, out, residual, pad_value)
    if p.w.dtype != torch.bfloat16 or p.stem:
        raise ValueError("conv2d (bf16 input): the layer's weights must be packed as bf16 (engine precision 'bf16'); no stem")
    ldx = _nhwc(x, "x", torch.bfloat16)
    n, h, w, cin = x.shape
    if cin != p.cin:
        raise ValueError(f"conv2d: input has {cin} channels, layer expects {p.cin}")
    ho = conv_out_size(h, p.kh, p.stride, p.pad, p.dil)
    wo = conv_out_size(w, p.kw, p.stride, p.pad, p.dil)
    if out is None:
        out = torch.empty((n, ho, wo, p.cout), dtype=torch.bfloat16, device=x.device)
    out_f32 = out.dtype == torch.float32
    ldy = _nhwc(out, "out", out.dtype)
    if tuple(out.shape) != (n, ho, wo, p.cout):
        raise ValueError(f"conv2d: out shape {tuple(out.shape)} != {(n, ho, wo, p.cout)}")
    ldr = 0
    if residual is not None:
        ldr = _nhwc(residual, "residual", torch.bfloat16)
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError("conv2d: residual shape mismatch")
    if pad_value is not None and (pad_value.numel() != cin or pad_value.dtype != torch.bfloat16 or not pad_value.is_contiguous()):
        raise ValueError(f"conv2d: pad_value must be a contiguous bf16 [{cin}] vector")
    shift = p.shift if shift_override is None else shift_override
    flags = (CONV_RELU if (p.relu if relu is None else relu) else 0) | (CONV_SHIFT_PER_IMAGE if per_image_shift else 0)

    def launch(t):
        d = ConvDesc(n, h, w, cin, ldx, ho, wo, p.cout, ldy, p.kh, p.kw, p.stride, p.pad, p.dil, ldr, p.kpad, flags, t)
        _lib.check(lib.pemp_conv2d_bf16_nhwc(C.byref(d), _p(x), _p(p.w), _p(out), _p(p.scale), _p(shift), _p(residual), _p(pad_value),
                                             1 if out_f32 else 0, _stream()), "pemp_conv2d_bf16_nhwc")

    if tile == 0:
        key = (p.cin, p.cout, p.kh, p.kw, p.stride, p.pad, p.dil, 5, n, h, w, int(residual is not None), int(pad_value is not None))   # 5: bf16
        tile = _TILE_CACHE.get(key)
        if tile is None:
            if _tunes(n * ho * wo):
                tile = _pick_tile(launch, p, key, p.cout, only=GROUP_TILES)
            else:
                tile = 24 if p.cout % 128 == 0 else 23
    launch(tile)
    return out


def convert(x, out):
    """Element-wise fp32 -> bf16 (round to nearest even) or bf16 -> fp32 between two contiguous tensors of one shape."""
    lib = _lib.load()
    _chk_dev(x, out)
    if not x.is_contiguous() or not out.is_contiguous() or x.numel() != out.numel() or x.numel() % 4:
        raise ValueError("convert: contiguous tensors of the same size (a multiple of 4 elements)")
    if x.dtype == torch.float32 and out.dtype == torch.bfloat16:
        _lib.check(lib.pemp_convert_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "convert_f32_bf16")
    elif x.dtype == torch.bfloat16 and out.dtype == torch.float32:
        _lib.check(lib.pemp_convert_bf16_f32(_p(x), _p(out), x.numel(), _stream()), "convert_bf16_f32")
    else:
        raise ValueError(f"convert: {x.dtype} -> {out.dtype} is not one of fp32 <-> bf16")
    return out


#: tile variants a grouped launch may use (conv_dma2.hip only)
GROUP_TILES = (23, 22, 25, 21, 24, 26, 27)
GROUP_MAX = 4


def _group_member_ok(x, p, pad_value):
    """What conv_dma2_supported (csrc/conv_dma2.hip) checks for one member of a grouped launch."""
    n, h, w, _ = x.shape
    ldx = _nhwc(x, "x")
    taps = p.kh * p.kw
    tapmax = (p.dil * (p.kh - 1) * w + p.dil * (p.kw - 1)) * ldx * 4
    xbytes = (n * h * w + p.pad * w + p.pad) * ldx * 4 + tapmax
    if p.stem or taps > 32 or xbytes >= 2 ** 31 or p.w.numel() * 4 >= 2 ** 31:
        return False
    if pad_value is not None:
        off = pad_value.data_ptr() - x.data_ptr()
        d = off + (p.pad * w + p.pad) * ldx * 4
        if off < n * h * w * ldx * 4 or d < tapmax or d + p.cin * 4 >= 2 ** 31:
            return False
    return True


def conv2d_group(xs, ps, outs, pad_values=None, residuals=None, tile=0):
    """Up to four INDEPENDENT convs in one launch (pemp_conv2d_group_nhwc_f32): member i computes ``outs[i] = act(scale *
    conv(xs[i], ps[i].w) + shift (+ residuals[i]))`` exactly as ``conv2d`` would -- bit-identical -- but the members' tiles share
    one grid.  ``pad_values``: per member or None (all or none).  The tile variant is timed once per group signature."""
    lib = _lib.load()
    n = len(ps)
    if not 1 <= n <= GROUP_MAX or len(xs) != n or len(outs) != n:
        raise ValueError(f"conv2d_group: 1..{GROUP_MAX} members with one input and one output each")
    pad_values = list(pad_values) if pad_values is not None else [None] * n
    residuals = list(residuals) if residuals is not None else [None] * n
    descs, keys = [], []
    for x, p, out, pv, res in zip(xs, ps, outs, pad_values, residuals):
        _chk_dev(x, p.w, out, pv, res)
        if p.stem:
            raise ValueError("conv2d_group: no stem convs")
        ldx, ldy = _nhwc(x, "x"), _nhwc(out, "out")
        nb, h, w, cin = x.shape
        if cin != p.cin:
            raise ValueError(f"conv2d_group: input has {cin} channels, layer expects {p.cin}")
        ho, wo = conv_out_size(h, p.kh, p.stride, p.pad, p.dil), conv_out_size(w, p.kw, p.stride, p.pad, p.dil)
        if tuple(out.shape) != (nb, ho, wo, p.cout):
            raise ValueError(f"conv2d_group: out shape {tuple(out.shape)} != {(nb, ho, wo, p.cout)}")
        ldr = 0
        if res is not None:
            ldr = _nhwc(res, "residual")
            if tuple(res.shape) != tuple(out.shape):
                raise ValueError("conv2d_group: residual shape mismatch")
        descs.append((nb, h, w, cin, ldx, ho, wo, p.cout, ldy, p.kh, p.kw, p.stride, p.pad, p.dil, ldr, p.kpad, CONV_RELU if p.relu else 0))
        keys += [p.cin, p.cout, p.kh, p.stride, p.pad, p.dil, nb, h, w, int(res is not None), int(pv is not None)]
    if not all(_group_member_ok(x, p, pv) for x, p, pv in zip(xs, ps, pad_values)):
        # a member outside the buffer-addressed kernels (tiny maps under a large dilation: the padding vector is not far enough
        # behind the activations; 2 GiB operands; > 32 taps): every member through its own launch -- same results
        for x, p, out, pv, res in zip(xs, ps, outs, pad_values, residuals):
            conv2d(x, p, out=out, residual=res, pad_value=pv if p.kh * p.kw > 1 else None)
        return outs
    arr = lambda ts: (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts])
    xa, wa, ya = arr(xs), arr([p.w for p in ps]), arr(outs)
    sa, ha, ra, pa = arr([p.scale for p in ps]), arr([p.shift for p in ps]), arr(residuals), arr(pad_values)
    ptr = lambda a: C.cast(a, C.POINTER(C.c_void_p))

    def launch(t):
        da = (ConvDesc * n)(*[ConvDesc(*d, t) for d in descs])
        _lib.check(lib.pemp_conv2d_group_nhwc_f32(n, da, ptr(xa), ptr(wa), ptr(ya), ptr(sa), ptr(ha), ptr(ra), ptr(pa), _stream()),
                   "pemp_conv2d_group_nhwc_f32")

    if tile == 0:
        key = (-7,) + tuple(keys)               # -7: a grouped launch (the cache file stores keys as integer lists)
        tile = _TILE_CACHE.get(key)
        if tile is None:
            if _tunes(max(d[0] * d[5] * d[6] for d in descs)):
                tile = _pick_tile(launch, None, key, min(p.cout for p in ps),
                                  only=[t for t in GROUP_TILES + (28,) if all(p.cout % TILE_VARIANTS[t][1] == 0 for p in ps)])
            else:
                tile = DEFAULT_TILE + 10
    launch(tile)
    return outs


#: split-K variants (ids 31..37 = the shapes of 21..27; pemp_hip.h): NOT bit-identical to the others.  The training convs may pick
#: them, and so may the evaluation path for SMALL row counts (one or two episodes per step: 5202 rows leave two thirds of the
#: chip idle on the 3x3 layers otherwise) -- see EVAL_SPLITK
SPLITK_TILES = (31, 32, 34, 35, 36, 37)
#: evaluation convs of at most EVAL_SPLITK_MAX_ROWS output rows may use the split-K variants.  OFF by default: every evaluation
#: variant is then bit-identical, a one-episode step equals the batched step bit for bit, and metrics cannot differ between
#: processes or ranks through the (timing-based) variant pick.  ON (PEMP_EVAL_SPLITK=1, ``with ops.eval_splitk():``,
#: ``Evaluator(splitk=True)``): one-episode steps are ~1.15x faster and agree with the exact path to rounding
#: (tests/test_eval_protocol_gpu.py states the bounds); the split-K hand-off uses device-scope write-through stores and sc1 loads
#: (csrc/conv_dma2.hip), not the fence pair of the HIP memory model.  Multi-rank jobs broadcast rank 0's picks (tuned_by_rank0).
EVAL_SPLITK = os.environ.get("PEMP_EVAL_SPLITK", "0") == "1"
EVAL_SPLITK_MAX_ROWS = int(os.environ.get("PEMP_EVAL_SPLITK_MAX_ROWS", "12000"))
#: Uncached split-K workspaces, one per (device, scope).  A workspace must never serve two launches that can run beside each
#: other, so everything that runs on its own stream has its own SCOPE: 0 = the main chain (training step, evaluation engine),
#: k = evaluation lane k (networks._HeadMixin.lane sets SK_SCOPE).  The size is fixed (every variant fits) and a workspace is
#: never freed or moved: captured hipGraphs hold its address.
_SK_WS = {}
SK_SCOPE = 0
_SK_WS_BYTES = 72 << 20            # 256 partial tiles of 256 x 256 floats + counters


class eval_splitk:
    """``with ops.eval_splitk(True):`` -- evaluation convs issued (or recorded into a hipGraph) inside the block may use the
    split-K variants for small row counts; restores the previous setting on exit."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global EVAL_SPLITK
        self.prev, EVAL_SPLITK = EVAL_SPLITK, self.on
        return self

    def __exit__(self, *exc):
        global EVAL_SPLITK
        EVAL_SPLITK = self.prev
        return False


def _splitk_ws(lib, desc, device):
    """-> (pointer, bytes) of the uncached workspace (pemp_uncached_alloc) of this device and the current scope."""
    need = lib.pemp_conv2d_splitk_workspace_bytes(C.byref(desc))
    if need == 0:
        return None, 0
    if need > _SK_WS_BYTES:
        raise RuntimeError(f"conv split-K: a launch asks for {need} bytes of workspace, the fixed size is {_SK_WS_BYTES}")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ent = _SK_WS.get((idx, SK_SCOPE))
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the conv split-K workspace must exist before a hipGraph is recorded: run the step eagerly once")
        with torch.cuda.device(device):
            ptr = lib.pemp_uncached_alloc(_SK_WS_BYTES)
        if not ptr:
            _lib.check(-1, "pemp_uncached_alloc")
        ent = _SK_WS[(idx, SK_SCOPE)] = (ptr, _SK_WS_BYTES)
    return ent


def _check_sk(lib, rc, ws, what):
    """A split-K launch that failed may have left arrival counters non-zero (no block would ever be "last" again): clear
    them before the failure is raised."""
    if rc and ws:
        lib.pemp_splitk_reset(C.c_void_p(ws), _stream())
    _lib.check(rc, what)


def _stats_rows(m, tile):
    """Partial rows the stats / bnbwd epilogues of variant ``tile`` write: one per row tile (pemp_conv2d_stats_rows)."""
    bm = TILE_VARIANTS[tile - 10 if tile > 30 else tile][0]
    return (m + bm - 1) // bm


def _train_tiles(cout):
    return [t for t in list(range(21, 28)) + list(SPLITK_TILES) if cout % TILE_VARIANTS[t - 10 if t > 30 else t][1] == 0]


def conv2d_stats(x, p, out=None, tile=0):
    """z = conv(x, w) with the per-32-row partial sums of z and z^2 left by the epilogue (pemp_conv2d_stats_nhwc_f32):
    -> (z, partials [row tiles of the chosen variant, 2, Cout]).  Raises PempHipError where the buffer-addressed kernels do not apply
    (callers then use conv2d + bn_stats); ``stats_supported`` says so beforehand."""
    lib = _lib.load()
    _chk_dev(x, p.w, out)
    ldx = _nhwc(x, "x")
    n, h, w, cin = x.shape
    if cin != p.cin:
        raise ValueError(f"conv2d_stats: input has {cin} channels, layer expects {p.cin}")
    if p.stem or p.scale is not None:
        raise ValueError("conv2d_stats: plain (non-stem, unscaled) convs only")
    ho = conv_out_size(h, p.kh, p.stride, p.pad, p.dil)
    wo = conv_out_size(w, p.kw, p.stride, p.pad, p.dil)
    if out is None:
        out = torch.empty((n, ho, wo, p.cout), dtype=torch.float32, device=x.device)
    ldy = _nhwc(out, "out")
    m = n * ho * wo
    part = torch.empty(((m + 63) // 64, 2, p.cout), dtype=torch.float32, device=x.device)     # the smallest row tile has 64 rows

    def launch(t):
        d = ConvDesc(n, h, w, cin, ldx, ho, wo, p.cout, ldy, p.kh, p.kw, p.stride, p.pad, p.dil, 0, p.kpad, 0, t)
        ws, ws_bytes = _splitk_ws(lib, d, x.device) if t > 30 else (None, 0)
        _check_sk(lib, lib.pemp_conv2d_stats_nhwc_f32(C.byref(d), _p(x), _p(p.w), _p(out), _p(part), C.c_void_p(ws), ws_bytes,
                                                      _stream()), ws, "pemp_conv2d_stats_nhwc_f32")

    if tile == 0:
        key = (p.cin, p.cout, p.kh, p.kw, p.stride, p.pad, p.dil, 2, n, h, w, 0, 0)     # 2: the stats epilogue
        tile = _TILE_CACHE.get(key)
        if tile is None:
            if _tunes(m):
                tile = _pick_tile(launch, p, key, p.cout, only=_train_tiles(p.cout) if SPLITK else range(21, 28))
            else:
                tile = DEFAULT_TILE + 10
    launch(tile)
    return out, part[:_stats_rows(m, tile)]


def conv2d_bnbwd(x, p, bn, residual=None, out=None, tile=0):
    """g = mask(conv(x, w) + residual) -- the gradient at the output of a train-mode BatchNorm(+ReLU), masked by that
    BatchNorm's ReLU -- with the per-32-row partial sums of g and g * xhat left by the epilogue
    (pemp_conv2d_bnbwd_nhwc_f32).  ``bn``: dict with z (the BatchNorm's input, NHWC like the result), mean, invstd and
    mask (int32 [M, C/32] sign bits from train_ops.bn_apply, or None for a BatchNorm without ReLU).
    -> (g, partials [row tiles of the chosen variant, 2, Cout])."""
    lib = _lib.load()
    z, mask = bn["z"], bn.get("mask")
    _chk_dev(x, p.w, out, residual, z, mask, bn["mean"], bn["invstd"])
    ldx = _nhwc(x, "x")
    n, h, w, cin = x.shape
    if cin != p.cin:
        raise ValueError(f"conv2d_bnbwd: input has {cin} channels, layer expects {p.cin}")
    if p.stem or p.scale is not None or p.shift is not None:
        raise ValueError("conv2d_bnbwd: plain (non-stem, no affine) convs only")
    ho = conv_out_size(h, p.kh, p.stride, p.pad, p.dil)
    wo = conv_out_size(w, p.kw, p.stride, p.pad, p.dil)
    if out is None:
        out = torch.empty((n, ho, wo, p.cout), dtype=torch.float32, device=x.device)
    ldy = _nhwc(out, "out")
    if tuple(z.shape) != tuple(out.shape):
        raise ValueError(f"conv2d_bnbwd: the BatchNorm input is {tuple(z.shape)}, the gradient {tuple(out.shape)}")
    ldz = _nhwc(z, "z")
    m = n * ho * wo
    if mask is not None and (mask.dtype != torch.int32 or not mask.is_contiguous() or mask.numel() != m * (p.cout // 32)):
        raise ValueError("conv2d_bnbwd: mask must be a contiguous int32 [M, Cout/32] tensor")
    ldr = 0
    if residual is not None:
        ldr = _nhwc(residual, "residual")
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError("conv2d_bnbwd: residual shape mismatch")
    part = torch.empty(((m + 63) // 64, 2, p.cout), dtype=torch.float32, device=x.device)

    def launch(t):
        d = ConvDesc(n, h, w, cin, ldx, ho, wo, p.cout, ldy, p.kh, p.kw, p.stride, p.pad, p.dil, ldr, p.kpad, 0, t)
        ws, ws_bytes = _splitk_ws(lib, d, x.device) if t > 30 else (None, 0)
        _check_sk(lib, lib.pemp_conv2d_bnbwd_nhwc_f32(C.byref(d), _p(x), _p(p.w), _p(out), _p(residual), _p(mask), _p(z), ldz,
                                                      _p(bn["mean"]), _p(bn["invstd"]), _p(part), C.c_void_p(ws), ws_bytes,
                                                      _stream()), ws, "pemp_conv2d_bnbwd_nhwc_f32")

    if tile == 0:
        key = (p.cin, p.cout, p.kh, p.kw, p.stride, p.pad, p.dil, 3, n, h, w, int(residual is not None), 0)   # 3: this epilogue
        tile = _TILE_CACHE.get(key)
        if tile is None:
            if _tunes(m):
                tile = _pick_tile(launch, p, key, p.cout, only=_train_tiles(p.cout) if SPLITK else range(21, 28))
            else:
                tile = DEFAULT_TILE + 10
    launch(tile)
    return out, part[:_stats_rows(m, tile)]


def dma2_supported(x, p):
    """Whether the buffer-addressed conv kernels apply to this (input, layer): what conv_dma2_supported checks in
    csrc/conv_dma2.hip (no padding value involved)."""
    n, h, w, cin = x.shape
    return (not p.stem and p.kh * p.kw <= 32 and cin % 32 == 0 and p.cout % 64 == 0
            and x.numel() * 4 < 2 ** 31 - (1 << 20) and p.w.numel() * 4 < 2 ** 31)


def stats_supported(x, p):
    """Whether conv2d_stats applies to this (input, layer)."""
    return p.scale is None and dma2_supported(x, p)


def fold_input_affine(p, s, t):
    """Fold a per-input-channel affine x -> s*x + t that sits IN FRONT of the (zero-padded) conv ``p`` into the conv
    itself (ASPPV2: BatchNorm before the dilated convs, networks/backbones.py:330-357).  Returns (ConvParams, pad_value):
    conv_W(s*x + t, pad 0) = conv_{W*s}(x, pad -t/s) + sum_taps W t, so out-of-image taps must read -t/s -- the value
    that is 0 in the affine's output space.  Needs s != 0 (returns None otherwise) and no output scale on ``p``."""
    if p.stem or p.scale is not None or bool((s == 0).any()) or not bool(torch.isfinite(t / s).all()):
        return None
    taps = p.kh * p.kw
    w = p.w[:, :taps * p.cin].view(p.cout, taps, p.cin)
    shift = (w.double() * t.double().view(1, 1, -1)).sum(dim=(1, 2)).float()
    if p.shift is not None:
        shift = shift + p.shift
    wf = (w * s.view(1, 1, -1)).reshape(p.cout, taps * p.cin).contiguous()
    q = ConvParams(wf, None, shift.contiguous(), p.cin, p.cout, p.kh, p.kw, p.stride, p.pad, p.dil, p.kpad, False, p.relu)
    return q, (-t / s).contiguous()


def pack_input(img_nchw, prior=None, out=None):
    """[N,3,H,W] (+ [N,1,H,W] prior) -> NHWC4."""
    lib = _lib.load()
    _chk_dev(img_nchw, prior)
    n, c, h, w = img_nchw.shape
    if c != 3 or img_nchw.dtype != torch.float32 or not img_nchw.is_contiguous():
        raise ValueError("pack_input: expected contiguous fp32 [N,3,H,W]")
    if prior is not None and (prior.dtype != torch.float32 or not prior.is_contiguous() or prior.numel() != n * h * w):
        raise ValueError("pack_input: prior must be contiguous fp32 [N,1,H,W]")
    if out is None:
        out = torch.empty((n, h, w, 4), dtype=torch.float32, device=img_nchw.device)
    _lib.check(lib.pemp_pack_input_nhwc4_f32(_p(img_nchw), _p(prior), _p(out), n, h, w, _stream()), "pack_input")
    return out


def _pool_out(i, k, s, p, ceil):
    num = i + 2 * p - k
    o = (-(-num // s) if ceil else num // s) + 1
    if ceil and (o - 1) * s >= i + p:
        o -= 1
    return o


def maxpool2d(x, k, s, p, ceil_mode=False, out=None):
    lib = _lib.load()
    _chk_dev(x)
    ldx = _nhwc(x, "x")
    n, h, w, c = x.shape
    ho, wo = _pool_out(h, k, s, p, ceil_mode), _pool_out(w, k, s, p, ceil_mode)
    if out is None:
        out = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
    ldy = _nhwc(out, "out")
    _lib.check(lib.pemp_maxpool2d_nhwc_f32(_p(x), _p(out), n, h, w, c, ldx, ho, wo, ldy, k, s, p, _stream()), "maxpool2d")
    return out


def global_avgpool(x, out=None):
    lib = _lib.load()
    _chk_dev(x)
    ldx = _nhwc(x, "x")
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _lib.check(lib.pemp_global_avgpool_nhwc_f32(_p(x), _p(out), n, h * w, c, ldx, _stream()), "global_avgpool")
    return out


def channel_affine_multi(x, scales, shifts, outs):
    """outs[b] = x * scales[b] + shifts[b] (per channel), up to 4 branches sharing one read of x."""
    lib = _lib.load()
    _chk_dev(x, *outs)
    nb = len(outs)
    if x.dim() == 2:
        m, c, ldx = x.shape[0], x.shape[1], x.stride(0)
        ldy = [o.stride(0) for o in outs]
    else:
        ldx = _nhwc(x, "x")
        m, c = x.shape[0] * x.shape[1] * x.shape[2], x.shape[3]
        ldy = [_nhwc(o, "out") for o in outs]
    arr = C.c_void_p * nb
    _lib.check(lib.pemp_channel_affine_multi_f32(
        _p(x), ldx, m, c, nb, arr(*[s.data_ptr() for s in scales]), arr(*[s.data_ptr() for s in shifts]),
        arr(*[o.data_ptr() for o in outs]), (C.c_int * nb)(*ldy), _stream()), "channel_affine_multi")
    return outs


def _ws(nbytes, device, cache=None, key=None):
    if cache is not None:
        t = cache.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            cache[key] = t
        return t
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)


def mpm_protos(sup_feat, sup_mask, ctr, B, S, p, ws_cache=None, out=None):
    """sup_feat [B*S,h,w,c] NHWC; sup_mask [B*S,2,H,W]; ctr [c,2p] -> protos [B,2p,c]."""
    lib = _lib.load()
    _chk_dev(sup_feat, sup_mask, ctr)
    ldf = _nhwc(sup_feat, "sup_feat")
    bs, h, w, c = sup_feat.shape
    H, W = sup_mask.shape[-2:]
    if bs != B * S or sup_mask.numel() != bs * 2 * H * W or not sup_mask.is_contiguous() or sup_mask.dtype != torch.float32:
        raise ValueError("mpm_protos: sup_mask must be contiguous fp32 [B*S,2,H,W]")
    if tuple(ctr.shape) != (c, 2 * p) or not ctr.is_contiguous():
        raise ValueError(f"mpm_protos: ctr must be contiguous [{c},{2 * p}]")
    nbytes = lib.pemp_mpm_workspace_bytes(B, S, h * w, c, p)
    ws = _ws(nbytes, sup_feat.device, ws_cache, ("mpm", B, S, h, w, c, p))
    if out is None:
        out = torch.empty((B, 2 * p, c), dtype=torch.float32, device=sup_feat.device)
    _lib.check(lib.pemp_mpm_protos_f32(_p(sup_feat), ldf, _p(sup_mask), _p(ctr), _p(out), _p(ws), ws.numel(),
                                       B, S, h, w, H, W, c, p, _stream()), "mpm_protos")
    return out


def masked_avg_pool(sup_feat, sup_mask, B, S, full_res, ws_cache=None, out=None):
    """-> protos [B,2,c] (row 0 fg, row 1 bg).  full_res=True is the Baseline form."""
    lib = _lib.load()
    _chk_dev(sup_feat, sup_mask)
    ldf = _nhwc(sup_feat, "sup_feat")
    bs, h, w, c = sup_feat.shape
    H, W = sup_mask.shape[-2:]
    if bs != B * S or sup_mask.numel() != bs * 2 * H * W or not sup_mask.is_contiguous() or sup_mask.dtype != torch.float32:
        raise ValueError("masked_avg_pool: sup_mask must be contiguous fp32 [B*S,2,H,W]")
    nbytes = lib.pemp_map_workspace_bytes(B, S, h * w, c)
    ws = _ws(nbytes, sup_feat.device, ws_cache, ("map", B, S, h, w, c))
    if out is None:
        out = torch.empty((B, 2, c), dtype=torch.float32, device=sup_feat.device)
    _lib.check(lib.pemp_masked_avg_pool_f32(_p(sup_feat), ldf, _p(sup_mask), _p(out), _p(ws), ws.numel(),
                                            B, S, h, w, H, W, c, 1 if full_res else 0, _stream()), "masked_avg_pool")
    return out


def cosine_proto_max(qry_feat, protos, dist_scalar, want_resp=False, pred=None, resp=None):
    """qry_feat [B,h,w,c]; protos [B,2p,c] -> pred [B,2,h,w] (+ resp uint8 [B,h,w])."""
    lib = _lib.load()
    _chk_dev(qry_feat, protos)
    ldf = _nhwc(qry_feat, "qry_feat")
    b, h, w, c = qry_feat.shape
    if protos.shape[0] != b or protos.shape[2] != c or not protos.is_contiguous():
        raise ValueError("cosine_proto_max: protos must be contiguous [B,2p,c]")
    p = protos.shape[1] // 2
    if pred is None:
        pred = torch.empty((b, 2, h, w), dtype=torch.float32, device=qry_feat.device)
    if want_resp and resp is None:
        resp = torch.empty((b, h, w), dtype=torch.uint8, device=qry_feat.device)
    _lib.check(lib.pemp_cosine_proto_max_f32(_p(qry_feat), ldf, _p(protos), _p(pred), _p(resp if want_resp else None),
                                             b, h * w, c, p, float(dist_scalar), _stream()), "cosine_proto_max")
    return (pred, resp) if want_resp else pred


def upsample_bilinear_ac(pred, out_hw):
    lib = _lib.load()
    _chk_dev(pred)
    b, c, h, w = pred.shape
    ho, wo = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=pred.device)
    _lib.check(lib.pemp_upsample_bilinear_ac_f32(_p(pred.contiguous()), _p(out), b, c, h, w, ho, wo, _stream()),
               "upsample_bilinear_ac")
    return out


def upsample_nearest_u8_i64(resp, out_hw):
    lib = _lib.load()
    _chk_dev(resp)
    b, h, w = resp.shape
    ho, wo = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((b, ho, wo), dtype=torch.int64, device=resp.device)
    _lib.check(lib.pemp_upsample_nearest_u8_i64(_p(resp.contiguous()), _p(out), b, h, w, ho, wo, _stream()),
               "upsample_nearest")
    return out


def cedt_weight(target, sigma=5.0, ws_cache=None):
    """CELossDT weight map (core/losses.py:23-41) on the device: target int64 [B,H,W] -> fp32 [B,H,W]."""
    lib = _lib.load()
    _chk_dev(target)
    if target.dtype != torch.int64 or not target.is_contiguous() or target.dim() != 3:
        raise ValueError("cedt_weight: target must be contiguous int64 [B,H,W]")
    b, h, w = target.shape
    out = torch.empty((b, h, w), dtype=torch.float32, device=target.device)
    ws = _ws(lib.pemp_cedt_workspace_bytes(b, h, w), target.device, ws_cache, ("cedt", b, h, w))
    _lib.check(lib.pemp_cedt_weight_f32(_p(target), _p(out), _p(ws), ws.numel(), b, h, w, float(sigma), _stream()), "cedt_weight")
    return out


def argmax_masks(pred):
    """pred [B,2,h,w] -> masks [B,2,h,w] fp32: channel 0 = (argmax == 1), channel 1 = (argmax == 0) (panet.py:169-171)."""
    lib = _lib.load()
    _chk_dev(pred)
    b, c, h, w = pred.shape
    if c != 2 or not pred.is_contiguous() or pred.dtype != torch.float32:
        raise ValueError("argmax_masks: pred must be contiguous fp32 [B,2,h,w]")
    masks = torch.empty_like(pred)
    _lib.check(lib.pemp_argmax_masks_f32(_p(pred), _p(masks), b, h * w, _stream()), "argmax_masks")
    return masks


def eval_tail(pred, target, want_logits=False, ws_cache=None, out_hw=None, weight=None):
    """pred [B,2,h,w]; target int64 [B,Ho,Wo] (or None with ``out_hw``: argmax only, statistics are zero)
    -> (argmax uint8 [B,Ho,Wo], stats f64 [B,8], logits|None)."""
    lib = _lib.load()
    _chk_dev(pred, target)
    b, c, h, w = pred.shape
    if c != 2 or not pred.is_contiguous():
        raise ValueError("eval_tail: pred must be contiguous [B,2,h,w]")
    if target is None:
        ho, wo = int(out_hw[0]), int(out_hw[1])
    else:
        if target.dtype != torch.int64 or not target.is_contiguous() or target.shape[0] != b:
            raise ValueError("eval_tail: target must be contiguous int64 [B,Ho,Wo]")
        ho, wo = target.shape[-2:]
    am = torch.empty((b, ho, wo), dtype=torch.uint8, device=pred.device)
    stats = torch.empty((b, 8), dtype=torch.float64, device=pred.device)
    logits = torch.empty((b, 2, ho, wo), dtype=torch.float32, device=pred.device) if want_logits else None
    nbytes = lib.pemp_eval_tail_workspace_bytes(b, ho, wo)
    ws = _ws(nbytes, pred.device, ws_cache, ("tail", b, ho, wo))
    if weight is not None and (weight.dtype != torch.float32 or not weight.is_contiguous() or tuple(weight.shape) != (b, ho, wo)):
        raise ValueError("eval_tail: weight must be contiguous fp32 [B,Ho,Wo]")
    _lib.check(lib.pemp_eval_tail_weighted_f32(_p(pred), _p(target), _p(weight), _p(am), _p(logits), _p(stats), _p(ws),
                                               ws.numel(), b, h, w, ho, wo, _stream()), "eval_tail")
    return am, stats, logits


def cm_reduce(x, mask_in, stride, want_argmax=False):
    """ResNetCM.comm statistics: x NHWC [N,h,w,C] (or None: pool the mask only); mask_in [N,Hm,Wm]
    -> (mask_out [N,h,w], stat [N,2,C] | None) (+ argmax int32 [N,C] with ``want_argmax``: the first maximal pixel,
    which train_ops.cm_bwd_add can use instead of searching for it again)."""
    lib = _lib.load()
    _chk_dev(x, mask_in)
    hm, wm = mask_in.shape[-2:]
    n = mask_in.shape[0]
    if x is None:
        h, w, c, ldx, stat = (hm + 2 - 3) // stride + 1, (wm + 2 - 3) // stride + 1, 0, 0, None
    else:
        ldx = _nhwc(x, "x")
        n, h, w, c = x.shape
        stat = torch.empty((n, 2, c), dtype=torch.float32, device=x.device)
    mask_out = torch.empty((n, h, w), dtype=torch.float32, device=mask_in.device)
    if want_argmax:
        arg = torch.empty((n, c), dtype=torch.int32, device=x.device)
        _lib.check(lib.pemp_cm_reduce_arg_f32(_p(x), ldx, _p(mask_in.contiguous()), _p(mask_out), _p(stat), _p(arg), n, hm, wm,
                                              h, w, c, stride, _stream()), "cm_reduce_arg")
        return mask_out, stat, arg
    _lib.check(lib.pemp_cm_reduce_f32(_p(x), ldx, _p(mask_in.contiguous()), _p(mask_out), _p(stat), n, hm, wm, h, w, c,
                                      stride, _stream()), "cm_reduce")
    return mask_out, stat


def cm_linear(stat, group, lin_w, lin_b, n_groups):
    """ResNetCM.comm after the statistics: stat [N,2,C] (or [N,2C]) -> (agg [G,2C] episode means, feat [G,2])."""
    lib = _lib.load()
    _chk_dev(stat, group, lin_w, lin_b)
    n = stat.shape[0]
    c2 = stat.numel() // n
    if group.dtype != torch.int32 or group.numel() != n or tuple(lin_w.shape) != (2, c2):
        raise ValueError("cm_linear: group must be int32 [N], lin_w [2, 2C]")
    agg = torch.empty((n_groups, c2), dtype=torch.float32, device=stat.device)
    feat = torch.empty((n_groups, 2), dtype=torch.float32, device=stat.device)
    _lib.check(lib.pemp_cm_linear_f32(_p(stat.contiguous()), _p(group), _p(lin_w.contiguous()), _p(lin_b.contiguous()), _p(agg),
                                      _p(feat), n, n_groups, c2, _stream()), "cm_linear")
    return agg, feat


def cm_bias(feat, group, wext, alpha=None, base=None):
    """Per-image bias of the two communication channels: [N, Cout] = base + alpha * (feat[group] @ wext^T).
    ``wext`` [Cout, 2] may be a strided column slice of the [Cout, C+2] weight matrix."""
    lib = _lib.load()
    _chk_dev(feat, group, wext, alpha, base)
    cout = wext.shape[0]
    if wext.dim() != 2 or wext.shape[1] != 2 or wext.stride(1) != 1:
        raise ValueError("cm_bias: wext must be [Cout, 2] with unit column stride")
    n = group.numel()
    out = torch.empty((n, cout), dtype=torch.float32, device=feat.device)
    _lib.check(lib.pemp_cm_bias_f32(_p(feat), _p(group), _p(wext), wext.stride(0), _p(alpha), _p(base), _p(out), n, cout,
                                    _stream()), "cm_bias")
    return out
