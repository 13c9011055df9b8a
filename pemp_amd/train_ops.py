"""Tensor-level wrappers over the training entry points of the C ABI (include/pemp_hip.h, "Training path").

Activations/gradients are NHWC fp32 views as in ``pemp_amd.ops``; 2-D ``[M, C]`` tensors are accepted
wherever the kernel only needs rows x channels.
"""
import ctypes as C

import torch

from . import _lib
from .ops import ConvDesc, CONV_STEM4, _p, _stream, _chk_dev, _ws, conv_out_size, _pool_out


def _rows(t, name):
    """-> (M, C, ld) of an NHWC view or a [M, C] matrix with unit channel stride."""
    if t.dtype != torch.float32 or t.stride(-1) != 1:
        raise ValueError(f"{name}: expected fp32 with unit channel stride")
    if t.dim() == 2:
        return t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else t.shape[1])
    if t.dim() == 4:
        from .ops import _nhwc
        n, h, w, c = t.shape
        return n * h * w, c, _nhwc(t, name)
    raise ValueError(f"{name}: expected 2-D or NHWC 4-D tensor")


def bn_stats(z, eps=1e-5, momentum=0.1, run_mean=None, run_var=None, ws_cache=None):
    """Batch mean and 1/sqrt(var+eps) per channel; updates running stats in place when given."""
    lib = _lib.load()
    _chk_dev(z, run_mean, run_var)
    m, c, ld = _rows(z, "z")
    mean = torch.empty(c, dtype=torch.float32, device=z.device)
    invstd = torch.empty(c, dtype=torch.float32, device=z.device)
    ws = _ws(lib.pemp_colsum_workspace_bytes(m, c), z.device, ws_cache, ("colsum", m, c))
    _lib.check(lib.pemp_bn_stats_f32(_p(z), ld, m, c, eps, momentum, _p(mean), _p(invstd), _p(run_mean), _p(run_var),
                                     _p(ws), ws.numel(), _stream()), "bn_stats")
    return mean, invstd


def _chk_part(part, what):
    if part.dtype != torch.float32 or not part.is_contiguous() or part.dim() != 3 or part.shape[1] != 2:
        raise ValueError(f"{what}: partials must be contiguous fp32 [row tiles, 2, C], got {tuple(part.shape)}")


def bn_stats_partials(part, m, eps=1e-5, momentum=0.1, run_mean=None, run_var=None):
    """bn_stats from the [row tiles, 2, C] partial sums a conv epilogue left (ops.conv2d_stats)."""
    lib = _lib.load()
    _chk_dev(part, run_mean, run_var)
    _chk_part(part, "bn_stats_partials")
    c = part.shape[2]
    mean = torch.empty(c, dtype=torch.float32, device=part.device)
    invstd = torch.empty(c, dtype=torch.float32, device=part.device)
    _lib.check(lib.pemp_bn_stats_partials_f32(_p(part), part.shape[0], m, c, eps, momentum, _p(mean), _p(invstd), _p(run_mean),
                                              _p(run_var), _stream()), "bn_stats_partials")
    return mean, invstd


def bn_fwd_partials(z, part, gamma, beta, out, eps=1e-5, momentum=0.1, run_mean=None, run_var=None, residual=None, relu=True,
                    mask=None):
    """bn_stats_partials + bn_apply in one call (pemp_bn_fwd_partials_f32): -> (out, mean, invstd)."""
    lib = _lib.load()
    _chk_dev(z, part, out, residual, mask, run_mean, run_var)
    _chk_part(part, "bn_fwd_partials")
    m, c, ldz = _rows(z, "z")
    _, _, ldy = _rows(out, "out")
    ldr = _rows(residual, "residual")[2] if residual is not None else 0
    if part.shape[2] != c:
        raise ValueError(f"bn_fwd_partials: partials have {part.shape[2]} channels, z has {c}")
    if mask is not None and (mask.dtype != torch.int32 or not mask.is_contiguous() or mask.numel() != m * (c // 32) or c % 32):
        raise ValueError(f"bn_fwd_partials: mask must be a contiguous int32 [{m}, {c}/32] tensor")
    mean = torch.empty(c, dtype=torch.float32, device=z.device)
    invstd = torch.empty(c, dtype=torch.float32, device=z.device)
    _lib.check(lib.pemp_bn_fwd_partials_f32(_p(z), ldz, _p(part), part.shape[0], m, c, eps, momentum, _p(gamma), _p(beta),
                                            _p(residual), ldr, _p(out), ldy, 1 if relu else 0, _p(mask), _p(mean), _p(invstd),
                                            _p(run_mean), _p(run_var), _stream()), "bn_fwd_partials")
    return out, mean, invstd


def bn_apply(z, mean, invstd, gamma, beta, out, residual=None, relu=True, mask=None):
    """``mask``: int32 [M, C/32] tensor that receives the sign bits of the result (bit c % 32 of mask[m, c // 32])."""
    lib = _lib.load()
    _chk_dev(z, out, residual, mask)
    m, c, ldz = _rows(z, "z")
    _, _, ldy = _rows(out, "out")
    ldr = _rows(residual, "residual")[2] if residual is not None else 0
    if mask is not None and (mask.dtype != torch.int32 or not mask.is_contiguous() or mask.numel() != m * (c // 32) or c % 32):
        raise ValueError(f"bn_apply: mask must be a contiguous int32 [{m}, {c}/32] tensor")
    _lib.check(lib.pemp_bn_apply_mask_f32(_p(z), ldz, _p(mean), _p(invstd), _p(gamma), _p(beta), _p(residual), ldr,
                                          _p(out), ldy, m, c, 1 if relu else 0, _p(mask), _stream()), "bn_apply")
    return out


def bn_apply_dropblock(z, mean, invstd, gamma, beta, out, dropblock):
    """BatchNorm apply (no residual, no ReLU) + the DropBlock2D behind it, one pass (pemp_bn_apply_dropblock_f32).
    ``dropblock`` = (mask fp32 [pixels], kept count int32 [1]) of ``dropblock_mask``; same arithmetic as bn_apply + pixel_scale."""
    lib = _lib.load()
    dmask, dcnt = dropblock
    _chk_dev(z, out, dmask, dcnt)
    m, c, ldz = _rows(z, "z")
    _, _, ldy = _rows(out, "out")
    if dmask.numel() != m or dmask.dtype != torch.float32 or not dmask.is_contiguous() or dcnt.dtype != torch.int32:
        raise ValueError("bn_apply_dropblock: mask must be a contiguous fp32 tensor with one entry per pixel, the count int32")
    _lib.check(lib.pemp_bn_apply_dropblock_f32(_p(z), ldz, _p(mean), _p(invstd), _p(gamma), _p(beta), _p(out), ldy, m, c,
                                               _p(dmask), _p(dcnt), _stream()), "bn_apply_dropblock")
    return out


def mask_supported(c):
    return c in (32, 64, 128, 256, 512, 1024)


def bn_bwd_partials(g, z, mean, invstd, gamma, part, dz, out=None):
    """Second half of the BatchNorm backward after ops.conv2d_bnbwd: -> (dgamma, dbeta); writes dz."""
    lib = _lib.load()
    _chk_dev(g, z, dz, part)
    m, c, ldg = _rows(g, "g")
    ldz = _rows(z, "z")[2]
    lddz = _rows(dz, "dz")[2]
    _chk_part(part, "bn_bwd_partials")
    if part.shape[2] != c:
        raise ValueError(f"bn_bwd_partials: partials have {part.shape[2]} channels, g has {c}")
    if out is None:
        dgamma = torch.empty(c, dtype=torch.float32, device=g.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=g.device)
    else:
        dgamma, dbeta = out
    _lib.check(lib.pemp_bn_bwd_partials_f32(_p(g), ldg, _p(z), ldz, _p(mean), _p(invstd), _p(gamma), _p(part), part.shape[0],
                                            _p(dz), lddz, _p(dgamma), _p(dbeta), m, c, _stream()), "bn_bwd_partials")
    return dgamma, dbeta


def bn_bwd(dy, y, z, mean, invstd, gamma, dz, gout=None, relu=True, ws_cache=None, out=None, mask=None):
    """-> (dgamma, dbeta); writes dz (and gout = dy*(y>0), the gradient of the residual branch).
    ``out=(dgamma, dbeta)``: contiguous [C] tensors to write into (e.g. the flat-buffer gradient views).
    ``mask``: the sign bits of y from bn_apply(mask=...); y is then not read."""
    lib = _lib.load()
    _chk_dev(dy, y, z, dz, gout, mask)
    m, c, lddy = _rows(dy, "dy")
    ldy = _rows(y, "y")[2] if y is not None else 0
    ldz = _rows(z, "z")[2]
    lddz = _rows(dz, "dz")[2]
    ldg = _rows(gout, "gout")[2] if gout is not None else 0
    if out is not None:
        dgamma, dbeta = out
        if not (dgamma.is_contiguous() and dbeta.is_contiguous() and dgamma.numel() == c and dbeta.numel() == c):
            raise ValueError("bn_bwd: out tensors must be contiguous [C]")
    else:
        dgamma = torch.empty(c, dtype=torch.float32, device=dy.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=dy.device)
    ws = _ws(lib.pemp_colsum_workspace_bytes(m, c), dy.device, ws_cache, ("colsum", m, c))
    if mask is not None and (mask.dtype != torch.int32 or not mask.is_contiguous() or mask.numel() != m * (c // 32) or c % 32):
        raise ValueError(f"bn_bwd: mask must be a contiguous int32 [{m}, {c}/32] tensor")
    _lib.check(lib.pemp_bn_bwd_mask_f32(_p(dy), lddy, _p(y), ldy, _p(mask if relu else None), _p(z), ldz, _p(mean), _p(invstd),
                                        _p(gamma), _p(dz), lddz, _p(gout), ldg, _p(dgamma), _p(dbeta), m, c, 1 if relu else 0,
                                        _p(ws), ws.numel(), _stream()), "bn_bwd")
    return dgamma, dbeta


def relu_bias_bwd(dy, y, g, add=None, relu=True, want_dbias=True, ws_cache=None, out=None):
    """g = (dy (+ add)) * (y > 0 if relu); returns dbias = column sums of g (or None); ``out``: contiguous [C]
    tensor to write dbias into (e.g. the bias gradient view of the flat buffer)."""
    lib = _lib.load()
    _chk_dev(dy, y, g, add)
    m, c, lddy = _rows(dy, "dy")
    ldy = _rows(y, "y")[2] if y is not None else 0
    lda = _rows(add, "add")[2] if add is not None else 0
    ldg = _rows(g, "g")[2]
    if out is not None and not (out.is_contiguous() and out.numel() == c):
        raise ValueError("relu_bias_bwd: out must be a contiguous [C] tensor")
    dbias = (out if out is not None else torch.empty(c, dtype=torch.float32, device=dy.device)) if want_dbias else None
    ws = _ws(lib.pemp_colsum_workspace_bytes(m, c), dy.device, ws_cache, ("colsum", m, c)) if want_dbias else None
    _lib.check(lib.pemp_relu_bias_bwd_f32(_p(dy), lddy, _p(y), ldy, _p(add), lda, _p(g), ldg, _p(dbias), m, c,
                                          1 if relu else 0, _p(ws), ws.numel() if ws is not None else 0, _stream()),
               "relu_bias_bwd")
    return dbias


from .ops import WGRAD_PICKS as _WGRAD_BLOCKS      # (layer geometry, input shape) -> (tile kind, block count) that measured fastest
WGRAD_BLOCK_CHOICES = (512, 768, 1024, 1536)


def conv_wgrad(x, g, p, dw, accumulate=False, ws_cache=None, variant=0, blocks=None):
    """dw (KRSC [Cout, Kpad_w]) (+)= wgrad of the conv described by ConvParams ``p`` (geometry only).
    For the stem ``dw`` has row length ceil(KH*KW*4 / 64) * 64.  ``variant`` = 1: the first-generation kernels
    (pointer-addressed; the library picks the buffer-addressed second generation where it applies, bit-identical).
    ``blocks``: how many blocks the pixel rows are split over (0: the library's 768; None: time WGRAD_BLOCK_CHOICES -- and,
    where both apply, the 128 x 128 and 64 x 64 tile kernels -- once per layer shape and keep the fastest; results of
    different splits differ by the rounding of the regrouped sum)."""
    lib = _lib.load()
    _chk_dev(x, g, dw)
    from .ops import _nhwc
    n, h, w, cin = x.shape
    ldx = _nhwc(x, "x")
    ho = conv_out_size(h, p.kh, p.stride, p.pad, p.dil)
    wo = conv_out_size(w, p.kw, p.stride, p.pad, p.dil)
    m, cout, ldg = _rows(g, "g")
    if m != n * ho * wo or cout != p.cout:
        raise ValueError(f"conv_wgrad: gradient has shape ({m},{cout}), expected ({n * ho * wo},{p.cout})")
    kpad = dw.shape[1]
    if dw.shape[0] != cout or not dw.is_contiguous():
        raise ValueError("conv_wgrad: dw must be contiguous [Cout, Kpad]")
    def launch(nb):          # nb: block count, or (tile kind 2 / 3, block count)
        kind, nb = nb if isinstance(nb, tuple) else (int(variant), nb)
        if variant:
            kind = int(variant)              # an explicit kernel generation wins over a remembered tile kind
        d = ConvDesc(n, h, w, cin, ldx, ho, wo, cout, ldg, p.kh, p.kw, p.stride, p.pad, p.dil, 0, kpad,
                     CONV_STEM4 if p.stem else 0, int(kind) | (int(nb) << 8))
        nbytes = lib.pemp_conv2d_wgrad_workspace_bytes(C.byref(d))
        ws = _ws(nbytes, x.device, ws_cache, ("wgrad", nbytes))
        _lib.check(lib.pemp_conv2d_wgrad_nhwc_f32(C.byref(d), _p(x), _p(g), _p(dw), 1 if accumulate else 0, _p(ws),
                                                  ws.numel(), _stream()), "conv_wgrad")

    if blocks is None:
        from . import ops
        # the pick depends on which kernel generation runs: the explicit ``variant`` and what decides the library's own choice
        # (geometry, strides) are part of the key -- a pick made for variant 1 must never be replayed for variant 0
        key = (cin, cout, p.kh, p.kw, p.stride, p.pad, p.dil, bool(p.stem), n, h, w, int(variant), ldx, ldg)
        blocks = _WGRAD_BLOCKS.get(key)
        if blocks is None:
            blocks = 0
            cands = list(WGRAD_BLOCK_CHOICES)
            if variant == 0 and not p.stem and cin % 128 == 0 and cout % 128 == 0:      # both tile sizes apply
                cands = [(k, nb) for k in (2, 3) for nb in WGRAD_BLOCK_CHOICES]
            if ops.PICK_HOOK is not None and not accumulate and not torch.cuda.is_current_stream_capturing():
                blocks = ops.PICK_HOOK("wgrad", list(cands), key)
                if blocks not in cands:
                    raise ValueError(f"PICK_HOOK returned {blocks!r}, not one of {cands}")
            elif ops.AUTOTUNE and not accumulate and m >= 4096 and not torch.cuda.is_current_stream_capturing():
                def timed(nb, reps=3):
                    launch(nb)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        launch(nb)
                    e1.record()
                    e1.synchronize()
                    return e0.elapsed_time(e1)
                ms = {nb: timed(nb) for nb in cands}
                for nb in cands:
                    ms[nb] = min(ms[nb], timed(nb))
                blocks = min(ms, key=ms.get)
            _WGRAD_BLOCKS[key] = blocks
            ops.save_picks()
    launch(blocks)
    return dw


def dgrad_mirror_table(layers, device):
    """``layers``: [(offset in floats, Cout, taps, Cin)] of the conv weights inside a flat parameter buffer -> (device int32
    table [L][6], total tiles) for ``dgrad_mirror``."""
    rows, first = [], 0
    for off, cout, taps, cin in layers:
        tci = (cin + 31) // 32
        rows.append([off, cout, taps, cin, first, tci])
        first += taps * ((cout + 31) // 32) * tci
    return torch.tensor(rows, dtype=torch.int32, device=device).contiguous(), first


def dgrad_mirror(params, mirror, table, total_tiles):
    """mirror[off : off + n] = the input-gradient KRSC weight ([Cin][taps flipped][Cout]) of every conv layer in ``table``,
    whose forward KRSC weight sits at params[off : off + n]: one launch per training step."""
    lib = _lib.load()
    _chk_dev(params, mirror, table)
    if params.numel() != mirror.numel() or not (params.is_contiguous() and mirror.is_contiguous()) or table.dtype != torch.int32:
        raise ValueError("dgrad_mirror: flat contiguous buffers of equal length and an int32 table required")
    _lib.check(lib.pemp_dgrad_mirror_f32(_p(params), _p(mirror), _p(table), table.shape[0], int(total_tiles), _stream()), "dgrad_mirror")
    return mirror


def dgrad_weight(w_krsc, kh, kw):
    """KRSC forward weight [Cout, KH*KW*Cin] -> the KRSC weight of the input-gradient conv
    [Cin, KH*KW*Cout] (taps flipped, channels transposed)."""
    cout = w_krsc.shape[0]
    cin = w_krsc.shape[1] // (kh * kw)
    w = w_krsc.view(cout, kh, kw, cin)
    return w.flip(1, 2).permute(3, 1, 2, 0).contiguous().view(cin, kh * kw * cout)


def maxpool_idx(x, k, s, p, ceil_mode=False):
    """nn.MaxPool2d forward that also records the winners -> (y, idx uint8 [n,ho,wo,c])."""
    lib = _lib.load()
    _chk_dev(x)
    if not x.is_contiguous():
        raise ValueError("maxpool_idx: contiguous NHWC tensor required")
    n, h, w, c = x.shape
    ho, wo = _pool_out(h, k, s, p, ceil_mode), _pool_out(w, k, s, p, ceil_mode)
    y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    _lib.check(lib.pemp_maxpool2d_idx_nhwc_f32(_p(x), _p(y), _p(idx), n, h, w, c, ho, wo, k, s, p, _stream()), "maxpool_idx")
    return y, idx


def maxpool_idx_bwd(idx, dy, in_hw, k, s, p):
    lib = _lib.load()
    _chk_dev(idx, dy)
    n, ho, wo, c = dy.shape
    h, w = in_hw
    if not (idx.is_contiguous() and dy.is_contiguous()) or idx.shape != dy.shape or idx.dtype != torch.uint8:
        raise ValueError("maxpool_idx_bwd: contiguous NHWC dy and matching uint8 idx required")
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
    _lib.check(lib.pemp_maxpool2d_idx_bwd_nhwc_f32(_p(idx), _p(dy), _p(dx), n, h, w, c, ho, wo, k, s, p, _stream()), "maxpool_idx_bwd")
    return dx


def maxpool_bwd(x, dy, k, s, p):
    lib = _lib.load()
    _chk_dev(x, dy)
    n, h, w, c = x.shape
    _, ho, wo, _ = dy.shape
    if not (x.is_contiguous() and dy.is_contiguous()):
        raise ValueError("maxpool_bwd: contiguous NHWC tensors required")
    dx = torch.empty_like(x)
    _lib.check(lib.pemp_maxpool2d_bwd_nhwc_f32(_p(x), _p(dy), _p(dx), n, h, w, c, ho, wo, k, s, p, _stream()), "maxpool_bwd")
    return dx


def scatter_strided(src, out_hw, s):
    lib = _lib.load()
    _chk_dev(src)
    n, hs, ws_, c = src.shape
    h, w = out_hw
    if not src.is_contiguous():
        raise ValueError("scatter_strided: contiguous NHWC source required")
    dst = torch.empty((n, h, w, c), dtype=torch.float32, device=src.device)
    _lib.check(lib.pemp_scatter_strided_nhwc_f32(_p(src), _p(dst), n, h, w, hs, ws_, c, s, _stream()), "scatter_strided")
    return dst


def gap_bwd_add(v, dst):
    """dst[n,i,:] += v[n,:] / HW."""
    lib = _lib.load()
    _chk_dev(v, dst)
    n, h, w, c = dst.shape
    _lib.check(lib.pemp_gap_bwd_add_nhwc_f32(_p(v.contiguous()), _p(dst), dst.stride(2), n, h * w, c, _stream()), "gap_bwd_add")
    return dst


class RandomStream:
    """(seed, offset) of the Philox stream the regulariser kernels draw from; every mask advances the offset,
    so successive layers and steps see different numbers and a re-seeded run repeats exactly."""

    def __init__(self, seed=0, device=None):
        self.seed, self.offset = int(seed) & (2 ** 64 - 1), 0
        # device-side step counter (added to the offset as step << 20 inside the kernels): lets a captured
        # hipGraph, whose kernel arguments are frozen, draw new numbers on every replay
        self.step = torch.zeros(1, dtype=torch.int64, device=device) if device is not None else None

    def next(self):
        self.offset += 1
        return self.seed, self.offset

    def begin_step(self):
        """Call once per training step (outside any graph capture)."""
        self.offset = 0
        if self.step is not None:
            self.step += 1


def dropblock_mask(n, h, w, drop_prob, block_size, rs, device, uniforms=None):
    """-> (mask fp32 [n,h,w] in {0,1}, kept_count int32 [1]) of one DropBlock2D call."""
    lib = _lib.load()
    mask = torch.empty((n, h, w), dtype=torch.float32, device=device)
    cnt = torch.empty(1, dtype=torch.int32, device=device)
    seed, off = rs.next()
    _lib.check(lib.pemp_dropblock_mask_f32(_p(mask), _p(cnt), _p(uniforms), n, h, w, float(drop_prob), int(block_size),
                                           seed, off, _p(rs.step), _stream()), "dropblock_mask")
    return mask, cnt


def pixel_scale(x, mask, cnt, out=None):
    """((x * mask[pixel]) * numel(mask)) / sum(mask): DropBlock2D forward and backward."""
    lib = _lib.load()
    _chk_dev(x, mask, cnt)
    m, c, ldx = _rows(x, "x")
    if mask.numel() != m:
        raise ValueError("pixel_scale: mask has a different pixel count")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(lib.pemp_pixel_scale_f32(_p(x), ldx, _p(mask), _p(cnt), _p(out), _rows(out, "out")[2], m, c, _stream()),
               "pixel_scale")
    return out


def dropout2d_mask(n, c, p, rs, device, uniforms=None):
    lib = _lib.load()
    mask = torch.empty((n, c), dtype=torch.float32, device=device)
    seed, off = rs.next()
    _lib.check(lib.pemp_dropout2d_mask_f32(_p(mask), _p(uniforms), n, c, float(p), seed, off, _p(rs.step), _stream()),
               "dropout2d_mask")
    return mask


def channel_scale(x, mask, out=None):
    """x[n,h,w,c] * mask[n,c]: Dropout2d forward and backward."""
    lib = _lib.load()
    _chk_dev(x, mask)
    from .ops import _nhwc
    n, h, w, c = x.shape
    if tuple(mask.shape) != (n, c):
        raise ValueError("channel_scale: mask must be [N, C]")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(lib.pemp_channel_scale_f32(_p(x), _nhwc(x, "x"), _p(mask), _p(out), _nhwc(out, "out"), n, h * w, c, _stream()),
               "channel_scale")
    return out


def cm_bias_bwd(colsum, feat, group, wext, dwext, dfeat_img, accumulate):
    """Backward of ``ops.cm_bias`` (training form, alpha = 1): writes dwext ([Cout,2] strided view of the weight
    gradient) and sets / adds to dfeat_img [N,2]."""
    lib = _lib.load()
    _chk_dev(colsum, feat, group, wext, dwext, dfeat_img)
    n, cout = colsum.shape
    if not colsum.is_contiguous() or wext.stride(1) != 1 or dwext.stride(1) != 1 or tuple(dfeat_img.shape) != (n, 2):
        raise ValueError("cm_bias_bwd: bad layouts")
    _lib.check(lib.pemp_cm_bias_bwd_f32(_p(colsum), _p(feat), _p(group), _p(wext), wext.stride(0), _p(dwext), dwext.stride(0),
                                        _p(dfeat_img), 1 if accumulate else 0, n, cout, _stream()), "cm_bias_bwd")


def cm_linear_bwd(dfeat_img, group, agg, lin_w, dlin_w, dlin_b):
    """Backward of ``ops.cm_linear``: writes dlin_w [2,2C], dlin_b [2]; returns dstat [N,2C]."""
    lib = _lib.load()
    _chk_dev(dfeat_img, group, agg, lin_w, dlin_w, dlin_b)
    n, (g, c2) = group.numel(), agg.shape
    if not (dlin_w.is_contiguous() and lin_w.is_contiguous() and agg.is_contiguous() and dfeat_img.is_contiguous()):
        raise ValueError("cm_linear_bwd: contiguous tensors required")
    dstat = torch.empty((n, c2), dtype=torch.float32, device=agg.device)
    _lib.check(lib.pemp_cm_linear_bwd_f32(_p(dfeat_img), _p(group), _p(agg), _p(lin_w), _p(dlin_w), _p(dlin_b), _p(dstat),
                                          n, g, c2, _stream()), "cm_linear_bwd")
    return dstat


def cm_bwd_add(x, mask, dstat, dx, argmax=None):
    """Backward of ``ops.cm_reduce``'s statistics: dx += mask * (dmean/HW + onehot(argmax) * dmax).
    x, dx: NHWC [N,h,w,C]; mask: the pooled mask cm_reduce returned [N,h,w]; dstat [N,2,C]; ``argmax`` int32 [N,C]
    from ``cm_reduce(want_argmax=True)`` skips the search for the maximal pixel."""
    lib = _lib.load()
    _chk_dev(x, mask, dstat, dx, argmax)
    from .ops import _nhwc
    n, h, w, c = x.shape
    if tuple(dx.shape) != (n, h, w, c) or tuple(dstat.shape) != (n, 2, c) or mask.numel() != n * h * w:
        raise ValueError("cm_bwd_add: shape mismatch")
    if argmax is not None:
        if argmax.dtype != torch.int32 or tuple(argmax.shape) != (n, c) or not argmax.is_contiguous():
            raise ValueError("cm_bwd_add: argmax must be contiguous int32 [N,C]")
        _lib.check(lib.pemp_cm_bwd_add_arg_f32(_p(mask.contiguous()), _p(dstat.contiguous()), _p(argmax), _p(dx),
                                               _nhwc(dx, "dx"), n, h * w, c, _stream()), "cm_bwd_add_arg")
        return dx
    _lib.check(lib.pemp_cm_bwd_add_f32(_p(x), _nhwc(x, "x"), _p(mask.contiguous()), _p(dstat.contiguous()), _p(dx),
                                       _nhwc(dx, "dx"), n, h * w, c, _stream()), "cm_bwd_add")
    return dx


def sgd_clip_step(params, grads, buf, max_norm, lr, momentum, weight_decay, first_step, grad_scale=1.0,
                  ws_cache=None, nesterov=False):
    """Flat-buffer clip_grad_norm_ + SGD step.  Returns the 1-element tensor holding ||grads||_2."""
    lib = _lib.load()
    _chk_dev(params, grads, buf)
    n = params.numel()
    if grads.numel() != n or buf.numel() != n or not (params.is_contiguous() and grads.is_contiguous() and buf.is_contiguous()):
        raise ValueError("sgd_clip_step: flat contiguous buffers of equal length required")
    norm = torch.empty(1, dtype=torch.float32, device=params.device)
    ws = _ws(lib.pemp_sgd_workspace_bytes(), params.device, ws_cache, ("sgd",))
    _lib.check(lib.pemp_sgd_clip_step_f32(_p(params), _p(grads), _p(buf), n, float(max_norm), float(lr), float(momentum),
                                          float(weight_decay), 1 if first_step else 0, float(grad_scale), 1 if nesterov else 0, _p(norm),
                                          _p(ws), ws.numel(), _stream()), "sgd_clip_step")
    return norm


def adam_clip_step(params, grads, exp_avg, exp_avg_sq, step, max_norm, lr, betas, eps, weight_decay, grad_scale=1.0, ws_cache=None):
    """Flat-buffer clip_grad_norm_ + torch.optim.Adam step (``step``: 1-based number of this update).  Returns the 1-element
    tensor holding ||grads||_2."""
    lib = _lib.load()
    _chk_dev(params, grads, exp_avg, exp_avg_sq)
    n = params.numel()
    if any(t.numel() != n or not t.is_contiguous() for t in (params, grads, exp_avg, exp_avg_sq)):
        raise ValueError("adam_clip_step: flat contiguous buffers of equal length required")
    norm = torch.empty(1, dtype=torch.float32, device=params.device)
    ws = _ws(lib.pemp_sgd_workspace_bytes(), params.device, ws_cache, ("sgd",))
    _lib.check(lib.pemp_adam_clip_step_f32(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), n, float(max_norm), float(lr),
                                           float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                           float(grad_scale), _p(norm), _p(ws), ws.numel(), _stream()), "adam_clip_step")
    return norm


def head_bwd_dlogits(sup_feat, qry_feat, sup_mask, ctr, fwd_ws, protos, dlogits, dfeat, B, S, p, dist_scalar,
                     ws_cache=None, map_full_res=False):
    """``head_bwd`` for an arbitrary gradient of the logits ``dlogits`` [B,2,Ho,Wo] (autograd's grad_output)."""
    lib = _lib.load()
    _chk_dev(sup_feat, qry_feat, sup_mask, ctr, fwd_ws, protos, dlogits, dfeat)
    from .ops import _nhwc
    ldf = _nhwc(sup_feat, "sup_feat")
    bs, h, w, c = sup_feat.shape
    H, W = sup_mask.shape[-2:]
    if dlogits.dtype != torch.float32 or not dlogits.is_contiguous() or dlogits.shape[:2] != (B, 2):
        raise ValueError("head_bwd_dlogits: dlogits must be contiguous fp32 [B,2,Ho,Wo]")
    ho, wo = dlogits.shape[-2:]
    ldd = _nhwc(dfeat, "dfeat")
    dctr = torch.empty((c, 2 * p), dtype=torch.float32, device=sup_feat.device) if p > 0 else None
    nbytes = lib.pemp_head_bwd_workspace_bytes(B, S, h * w, c, p)
    ws = _ws(nbytes, sup_feat.device, ws_cache, ("head_bwd", B, S, h, w, c, p))
    _lib.check(lib.pemp_head_bwd_dlogits_f32(_p(sup_feat), _p(qry_feat), ldf, _p(sup_mask), _p(ctr), _p(fwd_ws), _p(protos),
                                             _p(dlogits), _p(dfeat[:bs]), _p(dfeat[bs:]), ldd, _p(dctr), _p(ws), ws.numel(),
                                             B, S, h, w, H, W, ho, wo, c, p, 1 if map_full_res else 0, float(dist_scalar),
                                             _stream()), "head_bwd_dlogits")
    return dctr


def head_bwd(sup_feat, qry_feat, sup_mask, ctr, fwd_ws, protos, pred, target, stats, dfeat, B, S, p, dist_scalar,
             ws_cache=None, weight=None, map_full_res=False):
    """Gradient of the mean CE loss w.r.t. the features (written into ``dfeat`` [B*S + B, h, w, c], supports
    first) and w.r.t. ``ctr`` (returned, [c, 2p]; None for the plain-MAP head p == 0)."""
    lib = _lib.load()
    _chk_dev(sup_feat, qry_feat, sup_mask, ctr, fwd_ws, protos, pred, target, stats, dfeat)
    from .ops import _nhwc
    ldf = _nhwc(sup_feat, "sup_feat")
    bs, h, w, c = sup_feat.shape
    H, W = sup_mask.shape[-2:]
    ho, wo = target.shape[-2:]
    ldd = _nhwc(dfeat, "dfeat")
    dctr = torch.empty((c, 2 * p), dtype=torch.float32, device=sup_feat.device) if p > 0 else None
    nbytes = lib.pemp_head_bwd_workspace_bytes(B, S, h * w, c, p)
    ws = _ws(nbytes, sup_feat.device, ws_cache, ("head_bwd", B, S, h, w, c, p))
    _lib.check(lib.pemp_head_bwd_f32(_p(sup_feat), _p(qry_feat), ldf, _p(sup_mask), _p(ctr), _p(fwd_ws), _p(protos),
                                     _p(pred), _p(target), _p(weight), _p(stats), _p(dfeat[:bs]), _p(dfeat[bs:]), ldd, _p(dctr),
                                     _p(ws), ws.numel(), B, S, h, w, H, W, ho, wo, c, p, 1 if map_full_res else 0, float(dist_scalar),
                                     _stream()),
               "head_bwd")
    return dctr
