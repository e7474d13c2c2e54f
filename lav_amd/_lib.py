"""ctypes binding of liblav_amd.so - the FFI stub INTEGRATION.md describes.

The library is the product; this file only declares its C ABI (include/lav_amd.h).
There is no fallback: if the shared object is missing, cannot be loaded, or sees no
HIP device when an op is called, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LAV_AMD_LIB") or os.path.join(HERE, "liblav_amd.so")   # LAV_AMD_LIB: A/B a second build

ABI_VERSION = 28
MAX_CAM = 4


class Grid(C.Structure):
    _fields_ = [("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("ppm", C.c_float), ("nx", C.c_int), ("ny", C.c_int)]


class PointNet(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("num_input", C.c_int), ("channels", C.c_int)]


class Camera(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("l2w", C.c_float * 16), ("w2c", C.c_float * 16)]


class Conv(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "batch", "in_c_total", "in_c_offset", "cin", "h", "w", "cout", "kh", "kw", "stride", "pad_h", "pad_w",
        "dil_h", "dil_w", "transposed", "out_pad", "out_c_total", "out_c_offset", "relu_pre", "relu_post", "sigmoid", "target_cus")] + [("pad_value", C.c_float), ("precision", C.c_int)]


CONV_F32, CONV_BF16X6, CONV_F16X3 = 1, 2, 3     # lav_conv.precision (0 = library default: LAV_CONV_PRECISION, bf16x6)


# name -> (restype, argtypes); every symbol declared in include/lav_amd.h
_P, _I, _Z, _F = C.c_void_p, C.c_int, C.c_size_t, C.c_float
SIGNATURES = {
    "lav_abi_version": (_I, []),
    "lav_last_error": (C.c_char_p, []),
    "lav_device_count": (_I, []),
    "lav_profile_enable": (_I, [_I]),
    "lav_profile_reset": (_I, []),
    "lav_profile_read": (_I, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(_I)]),
    "lav_pillar_workspace_bytes": (_Z, [_I, _I, C.POINTER(Grid)]),
    "lav_pillar_workspace_init": (_I, [_P, _Z, _P]),
    "lav_pillar_decorate_workspace_bytes": (_Z, [_I, _I, C.POINTER(Grid)]),
    "lav_pillar_scatter": (_I, [_P, C.POINTER(_I), _I, _I, _I, C.POINTER(Grid), C.POINTER(PointNet), _P, _P, _P, _P,
                                _P, _Z, _P]),
    "lav_pillar_amax_count": (_I, [_I, C.POINTER(Grid)]),
    "lav_pillar_scatter_amax": (_I, [_P, C.POINTER(_I), _I, _I, _I, C.POINTER(Grid), C.POINTER(PointNet), _P, _P, _P, _P, _P,
                                     _P, _Z, _P]),
    "lav_paint": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, C.POINTER(Camera), _P, _P, _P]),
    "lav_gru_cast": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_gru_cast_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "lav_embed_cast": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "lav_gru_plan": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_gru_plan_steps": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_gru_plan_status": (_I, [_P, _Z, _I, _I, _I, _I, C.POINTER(_I), _P]),
    "lav_gru_plan_diag": (_I, [_P, _Z, _I, _I, _I, _I, C.POINTER(_I), _P]),
    "lav_gru_plan_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "lav_conv_out_hw": (_I, [C.POINTER(Conv), C.POINTER(_I), C.POINTER(_I)]),
    "lav_conv_packed_weight_floats": (_Z, [C.POINTER(Conv)]),
    "lav_conv_pack_weights": (_I, [C.POINTER(Conv), _P, _P]),
    "lav_conv_pack_map_ints": (_Z, [C.POINTER(Conv)]),
    "lav_conv_pack_map": (_I, [C.POINTER(Conv), _P]),
    "lav_conv_repack": (_I, [C.POINTER(Conv), _P, _P, _P, _P]),
    "lav_conv_repack_scratch": (_I, [C.POINTER(Conv), _P, _P, _P, _P, _Z, _P]),
    "lav_bn_fold": (_I, [_P, _P, _P, _P, C.c_double, _I, _P, _P, _P]),
    "lav_conv_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "lav_conv_wgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "lav_conv_wgrad_amax": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P, _I, _P, _I, _P]),
    "lav_absmax_parts": (_I, [_P, C.c_long, _P, _P]),
    "lav_conv_tile_info": (_I, [C.POINTER(Conv), C.POINTER(_I)]),
    "lav_conv_workspace_bytes": (_Z, [C.POINTER(Conv)]),
    "lav_conv2d": (_I, [C.POINTER(Conv), _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_conv_amax_count": (_I, [C.POINTER(Conv)]),
    "lav_conv2d_amax": (_I, [C.POINTER(Conv), _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _P, _P]),
    "lav_deconv_grouped": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I), _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "lav_crop_rotate": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _F, _I, _F, _F, _P, _P]),
    "lav_crop_rotate_indexed": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _I, _F, _I, _F, _F, _P, _P]),
    "lav_crop_rotate_backward": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _I, _F, _I, _F, _F, _P, _P]),
    "lav_pillar_decorate": (_I, [_P, C.POINTER(_I), _I, _I, _I, C.POINTER(Grid), _P, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_scatter_max": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "lav_scatter_max_backward": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "lav_conv1d_pair_packed_weight_floats": (_Z, [_I]),
    "lav_conv1d_pair_pack_weights": (_I, [_I, _P, _P]),
    "lav_upconv_pointwise_packed_floats": (_Z, [_I, _I, _I]),
    "lav_upconv_pointwise_pack": (_I, [_I, _I, _I, _P, _P]),
    "lav_upconv_pointwise_parts": (_I, [_I] * 8),
    "lav_upconv_pointwise": (_I, [_I] * 8 + [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "lav_conv1d_pair_chain_workspace_bytes": (_Z, [_I, _I]),
    "lav_conv1d_pair_chain_region": (_I, [_I, _I]),
    "lav_conv1d_pair_chain_lds_bytes": (_Z, [_I, _I, _I]),
    "lav_conv1d_pair_chain": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _P] + [C.POINTER(_P)] * 7 + [_P, _Z, _P]),
    "lav_conv1d_pair_chain_f16": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _P] + [C.POINTER(_P)] * 7 + [_P, _Z, _P]),
    "lav_conv1d_pair_chain_status": (_I, [_P, C.POINTER(_I), _P]),
    "lav_conv1d_pair_lds_bytes": (_Z, [_I, _I, _I]),
    "lav_conv1d_pair": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "lav_gru_seq_forward": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "lav_gru_seq_backward_workspace_bytes": (_Z, [_I, _I]),
    "lav_gru_seq_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "lav_bn_train_workspace_bytes": (_Z, [_I]),
    "lav_bn_train_forward": (_I, [_P, _P, _P, _I, _I, C.c_long, _P, _P, C.c_double, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "lav_bn_train_backward": (_I, [_P, _P, _P, _I, _I, C.c_long, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_bn_train_amax_count": (_I, [_I, _I, C.c_long]),
    "lav_bn_train_forward_amax": (_I, [_P, _P, _P, _I, _I, C.c_long, _P, _P, C.c_double, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_bn_train_backward_amax": (_I, [_P, _P, _P, _I, _I, C.c_long, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "lav_attn_pool": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "lav_linear_act": (_I, [_P, _I, _I, _P, _P, _I, _I, _P, _P]),
    "lav_maxpool3x3s2": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "lav_channel_affine": (_I, [_P, _I, _I, C.c_long, _P, _P, _P, _P]),
    "lav_nonfinite_count": (_I, [_I, C.POINTER(_P), C.POINTER(C.c_long), _P, _P]),
    "lav_copy_many": (_I, [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_Z), _P]),
    "lav_stage_many": (_I, [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I), C.POINTER(C.c_long), C.POINTER(_I), _P]),
    "lav_det_decode": (_I, [_P, _I, _I, _I] + [C.c_double] * 10 + [_P, _P, _P]),
    "lav_det_decode_report": (_I, [_P, _I, _I, _I] + [C.c_double] * 10 + [_P, _P, _P, _P, _P, _P]),
    "lav_stage_many_block": (_I, [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I), C.POINTER(C.c_long), C.POINTER(_I), _P, _I, _P, _P]),
    "lav_batch_limit": (_I, [_P]),
    "lav_pool_affine": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _P]),
    "lav_merge_ticks": (_I, [_P, _P, _I, _I, _P, _P]),
    "lav_stack_sweeps": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "lav_extract_peaks_workspace_bytes": (_Z, [_I, _I, _I]),
    "lav_extract_peaks": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _Z, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m lav_amd.build` (hipcc, gfx950). "
            "lav_amd has no CPU or pure-torch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.lav_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"liblav_amd ABI {got} != binding ABI {ABI_VERSION}: rebuild (python -m lav_amd.build --force)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().lav_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def require_device() -> None:
    n = load().lav_device_count()
    if n <= 0:
        msg = load().lav_last_error()
        raise RuntimeError(f"liblav_amd: no HIP device visible ({msg.decode() if msg else n}); "
                           "the LAV hot path has no CPU fallback")
