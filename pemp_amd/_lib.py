"""ctypes binding of libpemp_hip.so (the C ABI declared in include/pemp_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a symbol cannot be
resolved this module raises, loudly, at first use.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime our .so binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PEMP_HIP_LIB", os.path.join(_HERE, "libpemp_hip.so"))   # override: kernel experiments only

c_fp = C.c_void_p          # device pointers travel as integers
c_int = C.c_int
c_size = C.c_size_t


class ConvDesc(C.Structure):
    """struct pemp_conv_desc (include/pemp_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "Cin", "ldx", "Ho", "Wo", "Cout", "ldy", "KH", "KW",
                 "stride", "pad", "dil", "ldr", "Kpad")] + [("flags", C.c_uint32), ("tile", C.c_int32)]


class SampleDesc(C.Structure):
    """struct pemp_sample_desc (include/pemp_hip.h): one decoded sample of an episode."""
    _fields_ = [("img_off", C.c_int64), ("msk_off", C.c_int64), ("img_out", C.c_int64), ("msk_out", C.c_int64)] + \
               [(n, C.c_int32) for n in ("hs", "ws", "sh", "sw", "oy", "ox", "flip", "jitter_order")] + \
               [("jitter", C.c_float * 3), ("mask_mode", C.c_int32), ("ksx", C.c_int32), ("ksy", C.c_int32),
                ("ws_off", C.c_int64)]


CONV_RELU = 1
CONV_SHIFT_PER_IMAGE = 2
CONV_STEM4 = 4

#: every symbol include/pemp_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "pemp_last_error": (C.c_char_p, []),
    "pemp_abi_version": (c_int, []),
    "pemp_conv2d_hybrid_rows": (c_int, [C.POINTER(ConvDesc)]),
    "pemp_conv2d_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "pemp_conv2d_padv_splitk_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_size, c_fp]),
    "pemp_conv2d_padv_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "pemp_conv2d_group_nhwc_f32": (c_int, [c_int, C.POINTER(ConvDesc)] + [C.POINTER(c_fp)] * 7 + [c_fp]),
    "pemp_conv2d_bf16_nhwc": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "pemp_convert_f32_bf16": (c_int, [c_fp, c_fp, C.c_longlong, c_fp]),
    "pemp_convert_bf16_f32": (c_int, [c_fp, c_fp, C.c_longlong, c_fp]),
    "pemp_conv2d_dropblock_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_size, c_fp]),
    "pemp_bn_apply_dropblock_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp]),
    "pemp_pack_input_nhwc4_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "pemp_maxpool2d_nhwc_f32": (c_int, [c_fp, c_fp] + [c_int] * 11 + [c_fp]),
    "pemp_global_avgpool_nhwc_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "pemp_channel_affine_multi_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int,
                                              C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp),
                                              C.POINTER(c_int), c_fp]),
    "pemp_mpm_workspace_bytes": (c_size, [c_int] * 5),
    "pemp_mpm_protos_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_size] + [c_int] * 8 + [c_fp]),
    "pemp_map_workspace_bytes": (c_size, [c_int] * 4),
    "pemp_masked_avg_pool_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_size] + [c_int] * 8 + [c_fp]),
    "pemp_cosine_proto_max_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int,
                                          C.c_float, c_fp]),
    "pemp_upsample_bilinear_ac_f32": (c_int, [c_fp, c_fp] + [c_int] * 6 + [c_fp]),
    "pemp_upsample_nearest_u8_i64": (c_int, [c_fp, c_fp] + [c_int] * 5 + [c_fp]),
    "pemp_eval_tail_workspace_bytes": (c_size, [c_int] * 3),
    "pemp_eval_tail_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_size] + [c_int] * 5 + [c_fp]),
    "pemp_cm_reduce_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp] + [c_int] * 7 + [c_fp]),
    "pemp_argmax_masks_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_fp]),
    "pemp_cm_reduce_arg_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp] + [c_int] * 7 + [c_fp]),
    "pemp_cm_bwd_add_arg_f32": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 4 + [c_fp]),
    "pemp_cm_linear_f32": (c_int, [c_fp] * 6 + [c_int] * 3 + [c_fp]),
    "pemp_cm_bias_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
    "pemp_cm_bias_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_fp, c_int, c_int, c_int, c_fp]),
    "pemp_cm_linear_bwd_f32": (c_int, [c_fp] * 7 + [c_int] * 3 + [c_fp]),
    # episode input pipeline
    "pemp_episode_plan": (c_size, [C.POINTER(SampleDesc), c_int, c_int, c_int]),
    "pemp_episode_preprocess": (c_int, [c_fp, C.POINTER(SampleDesc), c_fp, c_int, c_int, c_int, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), c_fp, c_fp, c_fp, c_fp, c_size, c_fp]),
    # training path
    "pemp_dropblock_mask_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, C.c_float, c_int, C.c_uint64, C.c_uint64, c_fp,
                                        c_fp]),
    "pemp_pixel_scale_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_int, C.c_longlong, c_int, c_fp]),
    "pemp_dropout2d_mask_f32": (c_int, [c_fp, c_fp, c_int, c_int, C.c_float, C.c_uint64, C.c_uint64, c_fp, c_fp]),
    "pemp_channel_scale_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "pemp_cm_bwd_add_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "pemp_conv2d_splitk_workspace_bytes": (c_size, [C.POINTER(ConvDesc)]),
    "pemp_conv2d_splitk_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_size, c_fp]),
    "pemp_uncached_alloc": (c_fp, [c_size]),
    "pemp_uncached_free": (c_int, [c_fp]),
    "pemp_splitk_reset": (c_int, [c_fp, c_fp]),
    "pemp_spin_us": (c_int, [c_int, c_fp]),
    "pemp_conv2d_stats_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_size, c_fp]),
    "pemp_conv2d_stats_rows": (c_int, [C.POINTER(ConvDesc)]),
    "pemp_bn_stats_partials_f32": (c_int, [c_fp, c_int, c_int, c_int, C.c_float, C.c_float, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "pemp_bn_fwd_partials_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_int, c_int, C.c_float, C.c_float, c_fp, c_fp, c_fp, c_int,
                                         c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "pemp_conv2d_bnbwd_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp,
                                           c_size, c_fp]),
    "pemp_bn_bwd_partials_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_int,
                                         c_int, c_fp]),
    "pemp_bn_apply_mask_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_int, c_int,
                                       c_int, c_fp, c_fp]),
    "pemp_conv2d_wgrad_workspace_bytes": (c_size, [C.POINTER(ConvDesc)]),
    "pemp_conv2d_wgrad_nhwc_f32": (c_int, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp, c_int, c_fp, c_size, c_fp]),
    "pemp_colsum_workspace_bytes": (c_size, [c_int, c_int]),
    "pemp_bn_stats_f32": (c_int, [c_fp, c_int, c_int, c_int, C.c_float, C.c_float, c_fp, c_fp, c_fp, c_fp,
                                  c_fp, c_size, c_fp]),
    "pemp_bn_apply_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_int, c_int,
                                  c_int, c_fp]),
    "pemp_bn_bwd_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int,
                                c_fp, c_fp, c_int, c_int, c_int, c_fp, c_size, c_fp]),
    "pemp_bn_bwd_mask_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int,
                                     c_fp, c_fp, c_int, c_int, c_int, c_fp, c_size, c_fp]),
    "pemp_relu_bias_bwd_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_int, c_fp, c_int, c_fp, c_int, c_int,
                                       c_int, c_fp, c_size, c_fp]),
    "pemp_maxpool2d_bwd_nhwc_f32": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 9 + [c_fp]),
    "pemp_maxpool2d_idx_nhwc_f32": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 9 + [c_fp]),
    "pemp_maxpool2d_idx_bwd_nhwc_f32": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 9 + [c_fp]),
    "pemp_scatter_strided_nhwc_f32": (c_int, [c_fp, c_fp] + [c_int] * 7 + [c_fp]),
    "pemp_gap_bwd_add_nhwc_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "pemp_head_bwd_workspace_bytes": (c_size, [c_int] * 5),
    "pemp_head_bwd_f32": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int,
                                  c_fp, c_fp, c_size] + [c_int] * 11 + [C.c_float, c_fp]),
    "pemp_head_bwd_dlogits_f32": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int,
                                          c_fp, c_fp, c_size] + [c_int] * 11 + [C.c_float, c_fp]),
    "pemp_eval_tail_weighted_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_size] + [c_int] * 5 + [c_fp]),
    "pemp_cedt_workspace_bytes": (c_size, [c_int] * 3),
    "pemp_cedt_weight_f32": (c_int, [c_fp, c_fp, c_fp, c_size, c_int, c_int, c_int, C.c_float, c_fp]),
    "pemp_sgd_workspace_bytes": (c_size, []),
    "pemp_sgd_clip_step_f32": (c_int, [c_fp, c_fp, c_fp, C.c_longlong, C.c_float, C.c_float, C.c_float,
                                       C.c_float, c_int, C.c_float, c_int, c_fp, c_fp, c_size, c_fp]),
    "pemp_dgrad_mirror_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_fp]),
    "pemp_adam_clip_step_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, C.c_longlong, C.c_float] + [C.c_double] * 5 + [C.c_longlong, C.c_float, c_fp, c_fp,
                                                                                                  c_size, c_fp]),
}

ABI_VERSION = 2          # include/pemp_hip.h: PEMP_ABI_VERSION
_lib = None


class PempHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle with prototypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PempHipError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "pemp_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.pemp_abi_version() != ABI_VERSION:
        raise PempHipError(f"ABI version mismatch: library {lib.pemp_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pemp_last_error().decode(errors="replace")
        kind = "invalid argument" if rc < 0 else f"hipError {rc}"
        raise PempHipError(f"{what}: {kind}: {msg}")
