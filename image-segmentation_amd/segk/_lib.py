"""ctypes binding of libsegk.so (the C ABI declared in include/segk.h).

The library is required for every computation on GPU tensors: if it is missing, `lib()` raises --
there is no silent eager fallback for device tensors.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEGK_LIB_PATH: another build of the same library (kernel experiments: A/B two builds in one session)
LIB_PATH = os.environ.get("SEGK_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libsegk.so")

SEGK_F32, SEGK_BF16 = 0, 1
_ERR_NAMES = {-1: "SEGK_EINVAL", -2: "SEGK_ESHAPE", -3: "SEGK_EALIGN", -4: "SEGK_EWORKSPACE", -5: "SEGK_EHIP"}

P, I, L, F, Z = c_void_p, c_int, c_int64, c_float, c_size_t

# name -> (restype, argtypes); int-returning functions are status codes
_PROTOS = {
    "segk_last_error": (ctypes.c_char_p, []),
    "segk_version": (I, []),
    "segk_dice_focal_workspace": (Z, [L, L, L]),
    "segk_dice_focal_fwd": (I, [P, I, P, P, L, L, L, F, P, I, P, P, P, P, Z, I, P]),
    "segk_dice_focal_fwd_static": (I, [P, I, P, P, L, L, L, F, P, I, P, P, P, I, P]),
    "segk_dice_focal_finalize": (I, [P, P, L, L, L, I, P, I, P]),
    "segk_dice_focal_bwd": (I, [P, I, P, P, L, L, L, F, P, I, P, P, P, P, P, I, P]),
    "segk_argmax_dice": (I, [P, I, P, L, L, L, P, P, P, I, P]),
    "segk_edge_workspace": (Z, [L, L, L]),
    "segk_edge_bwd_workspace": (Z, [L, L, L]),
    "segk_edge_fwd": (I, [P, L, L, L, F, P, P, P, Z, I, P]),
    "segk_edge_bwd": (I, [P, P, P, L, L, L, P, P, Z, I, P]),
    "segk_nchw_to_nhwc": (I, [P, L, L, L, L, P, I, L, L, I, P]),
    "segk_im2col3x3": (I, [P, L, L, L, L, P, I, L, L, I, P]),
    "segk_conv_first_stat_rows": (L, [L, L, L]),
    "segk_conv_first_fwd": (I, [P, L, L, L, L, P, P, L, P, I, L, P, I, I, P]),
    "segk_conv_first_norm": (I, [I, P, L, L, L, L, P, P, L, P, P, P, P, P, L, P, L, P, I, P]),
    "segk_norm_bwd_sums_finalize": (I, [P, L, L, P, P, P, I, P, Z, I, P]),
    "segk_conv_first_norm_bwd_workspace": (Z, [L, L, L]),
    "segk_conv_first_norm_bwd": (I, [P, L, L, L, L, P, P, P, P, P, P, L, P, P, P, I, P, Z, I, P]),
    "segk_conv_first_wgrad_workspace": (Z, [L, L, L]),
    "segk_conv_first_wgrad": (I, [P, L, L, L, L, P, I, L, L, P, I, P, Z, I, P]),
    "segk_conv_krow": (I, [I]),
    "segk_conv_packed_bytes": (Z, [I, L, L, I, L]),
    "segk_conv_pack": (I, [P, L, L, I, L, I, P, P, I, P]),
    "segk_conv_pack_batch": (I, [I, P, L, L, I, P]),
    "segk_conv_fwd": (I, [I, I, P, L, L, P, L, L, L, L, L, P, L, P, P, L, P, L, L, P, L, L, P, I, I, P]),
    "segk_conv_stat_rows": (L, [L, L, L, L]),
    "segk_conv_stat_rows_fine": (L, [L, L, L]),
    "segk_conv_dgrad_norm": (I, [I, I, P, L, L, L, L, L, P, L, P, L, P, L, P, P, P, P, I, P]),
    "segk_norm_stats_finalize_workspace": (Z, [L, L, L]),
    "segk_norm_stats_finalize": (I, [P, L, L, L, L, F, P, P, P, F, P, Z, I, P]),
    "segk_conv_wgrad_workspace": (Z, [I, I, L, L, L, L, L, L]),
    "segk_conv_wgrad": (I, [I, I, P, L, L, P, L, L, L, L, L, P, L, L, L, P, I, P, Z, I, P]),
    "segk_convt_packed_bytes": (Z, [I, L, L]),
    "segk_convt_pack": (I, [P, L, L, I, P, P, I, P]),
    "segk_convt_fwd": (I, [I, P, L, L, L, L, L, P, L, P, P, L, I, I, P]),
    "segk_convt_dgrad": (I, [I, P, L, L, L, L, L, P, L, P, L, P, L, I, I, P]),
    "segk_convt_wgrad_workspace": (Z, [I, L, L, L, L, L]),
    "segk_convt_wgrad": (I, [I, P, L, L, L, L, L, P, L, L, P, I, P, Z, I, P]),
    "segk_conv3d_packed_bytes": (Z, [I, L, L]),
    "segk_conv3d_pack": (I, [P, L, L, I, P, P, I, P]),
    "segk_conv3d_fwd": (I, [I, P, L, L, L, L, L, L, P, L, P, L, I, I, P]),
    "segk_conv3d_wgrad_workspace": (Z, [I, L, L, L, L, L, L]),
    "segk_conv3d_wgrad": (I, [I, P, L, L, L, L, L, L, P, L, L, P, I, P, Z, I, P]),
    "segk_convd_packed_bytes": (Z, [I, L, L, L]),
    "segk_convd_pack": (I, [P, L, L, L, I, P, P, I, P]),
    "segk_convd_fwd": (I, [I, P, L, L, L, L, L, L, L, P, L, P, L, P, I, I, P]),
    "segk_convd_wgrad_workspace": (Z, [I, L, L, L, L, L, L, L]),
    "segk_convd_wgrad": (I, [I, P, L, L, L, L, L, L, L, P, L, L, L, P, I, P, Z, I, P]),
    "segk_im2col3d": (I, [I, P, L, L, L, L, L, L, P, P, P, P, L, L, I, P]),
    "segk_col2im3d": (I, [I, P, L, L, L, L, L, L, P, P, P, P, L, I, I, P]),
    "segk_maxpool3d_fwd": (I, [I, P, L, L, L, L, L, L, P, L, P, I, P]),
    "segk_maxpool3d_bwd": (I, [I, P, L, P, L, L, L, L, L, P, L, I, I, P]),
    "segk_avgpool3d_fwd": (I, [I, P, L, L, L, L, L, L, P, L, I, P]),
    "segk_avgpool3d_bwd": (I, [I, P, L, L, L, L, L, L, P, L, I, I, P]),
    "segk_rows_mean_fwd": (I, [I, P, L, L, L, L, P, I, P]),
    "segk_rows_mean_bwd": (I, [I, P, L, L, L, P, L, I, P]),
    "segk_linear_fwd": (I, [P, P, P, L, L, L, P, I, P]),
    "segk_linear_bwd": (I, [P, P, P, L, L, L, P, P, P, I, I, P]),
    "segk_relu_bwd": (I, [I, P, L, P, L, L, L, P, L, I, P]),
    "segk_head1x1_fwd": (I, [I, P, L, L, L, L, P, P, L, P, I, P]),
    "segk_head1x1_bwd_workspace": (Z, [L, L, L, L]),
    "segk_head1x1_bwd": (I, [I, P, L, L, L, L, P, L, P, P, L, I, P, P, P, Z, I, P]),
    "segk_norm_workspace": (Z, [L, L, L]),
    "segk_norm_stats": (I, [I, P, L, L, L, L, F, P, P, P, F, P, Z, I, P]),
    "segk_norm_apply": (I, [I, P, L, L, L, L, P, P, P, I, P, L, I, P]),
    "segk_norm_eval_stats": (I, [P, P, L, F, P, I, P]),
    "segk_norm_bwd": (I, [I, P, L, P, L, L, L, L, P, P, P, I, P, L, P, P, P, P, Z, I, P]),
    "segk_norm_bwd_finish_workspace": (Z, [L, L]),
    "segk_norm_bwd_finish": (I, [I, P, L, P, L, P, L, L, L, P, P, P, I, P, L, P, P, P, Z, I, P]),
    "segk_norm_pool_apply": (I, [I, P, L, L, L, L, L, P, P, P, P, L, P, L, I, P]),
    "segk_norm_pool_bwd": (I, [I, P, L, P, L, P, L, L, L, L, L, P, P, P, I, P, L, P, P, P, Z, I, P]),
    "segk_maxpool2x2_fwd": (I, [I, P, L, L, L, L, L, P, L, I, P]),
    "segk_maxpool2x2_bwd": (I, [I, P, L, P, L, L, L, L, L, P, L, I, I, P]),
    "segk_upsample2x_fwd": (I, [I, P, L, L, L, L, L, P, I, P, L, I, P]),
    "segk_upsample2x_bwd": (I, [I, P, L, P, L, L, L, L, L, I, P, L, I, P]),
    "segk_colsum_workspace": (Z, [L, L]),
    "segk_colsum": (I, [I, P, L, L, L, P, I, P, Z, I, P]),
    "segk_copy_slice": (I, [I, P, L, P, L, L, L, I, I, P]),
    "segk_block2": (I, [I, P, L, P, L, L, L, L, L, I, I, P]),
    "segk_resize_bilinear_fwd": (I, [I, P, L, L, L, L, L, L, L, F, F, P, I, P, L, I, P]),
    "segk_resize_bilinear_bwd": (I, [I, P, L, L, L, L, L, L, L, F, F, P, L, I, I, P]),
    "segk_adam_step": (I, [P, P, P, P, L, F, F, F, F, F, L, F, I, P]),
}

_lib = None
launches = 0   # number of C-ABI calls made by this process (tests assert the HIP path really ran)
calls = {}     # the same per entry point


class SegkError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_PROTOS)


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SegkError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C image-segmentation_amd/csrc`); "
                "GPU tensors are only ever computed by the HIP library")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def call(name, *args):
    """Invoke a status-returning entry point; raise SegkError with the library's message on failure."""
    global launches
    h = lib()
    rc = getattr(h, name)(*args)
    launches += 1
    calls[name] = calls.get(name, 0) + 1
    if rc != 0:
        msg = h.segk_last_error().decode(errors="replace")
        raise SegkError(f"{name} failed with {_ERR_NAMES.get(rc, rc)}: {msg}")


def query(name, *args):
    """Invoke a size/int query (no status code)."""
    return getattr(lib(), name)(*args)
