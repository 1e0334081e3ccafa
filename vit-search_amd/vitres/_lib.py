"""ctypes loader for libvitres_hip.so (the C ABI declared in include/vitres_hip.h).

The library is the product: there is no CPU / PyTorch fallback.  Importing this module never fails
(so that host-side logic can be unit-tested without a GPU), but the first kernel call raises
RuntimeError when the shared object is missing or does not export a declared symbol.
"""
import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITRES_LIB: another build of the same ABI (A/B timing of kernel changes on one box: tools/ab.sh)
LIB_PATH = os.environ.get("VITRES_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libvitres_hip.so")

VR_F32, VR_BF16 = 0, 1


class RowMap(ctypes.Structure):
    _fields_ = [("rpi", c_int32), ("rps", c_int32), ("off", c_int32)]


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p),
        ("bias", c_void_p), ("pos", c_void_p), ("scale", c_void_p), ("keep_n", c_void_p),
        ("resid", c_void_p), ("dact_u", c_void_p), ("keep_k", c_void_p), ("bias_grad", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("ldu", c_int32),
        ("a_trans", c_int32), ("b_trans", c_int32),
        ("in_dtype", c_int32), ("out_dtype", c_int32),
        ("act", c_int32), ("atomic", c_int32), ("split_k", c_int32), ("rows_in", c_int32), ("n_period", c_int32), ("k_period", c_int32), ("sched", c_int32),
        ("a_map", RowMap), ("b_map", RowMap), ("c_map", RowMap), ("m_groups", c_int32),
        ("ws", c_void_p), ("ws_bytes", c_int64), ("ring", c_int32), ("k_shares", c_int32),
    ]


class LnEpilogue(ctypes.Structure):
    _fields_ = [("mode", c_int32), ("eps", c_float), ("w", c_void_p), ("b", c_void_p), ("keep", c_void_p), ("y", c_void_p),
                ("mean", c_void_p), ("rstd", c_void_p), ("x", c_void_p), ("dw", c_void_p), ("db", c_void_p),
                ("gt_out", c_void_p), ("gt_scale", c_void_p), ("gt_keep", c_void_p), ("grad_copies", c_int32)]


MAX_ZERO_RANGES = 24


class ZeroRanges(ctypes.Structure):
    _fields_ = [("lo", c_int64 * MAX_ZERO_RANGES), ("count", c_int64 * MAX_ZERO_RANGES), ("n", c_int32), ("reserved", c_int32)]


class LnGradSlot(ctypes.Structure):
    _fields_ = [("part_w", c_void_p), ("part_b", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("C", c_int32),
                ("reserved", c_int32)]


# name -> argtypes ; every entry point returns int.  Must list every symbol of include/vitres_hip.h
SYMBOLS = {
    "vr_version": [],
    "vr_gemm": [ctypes.POINTER(GemmArgs), c_void_p],
    "vr_gemm_group": [ctypes.POINTER(GemmArgs), ctypes.c_int32, c_void_p],
    "vr_gemm_ln": [ctypes.POINTER(GemmArgs), ctypes.POINTER(LnEpilogue), c_void_p],
    "vr_gemm_ln_supported": [c_int32],
    "vr_gemm_ln_fold": [ctypes.POINTER(GemmArgs), ctypes.POINTER(LnEpilogue), c_void_p],
    "vr_gemm_ws_bytes": [],
    "vr_cast_f32_bf16": [c_void_p, c_void_p, c_int64, c_void_p],
    "vr_adamw_flat": [c_void_p] * 6 + [c_float, c_void_p, c_void_p, c_int32, c_int64, c_void_p],
    "vr_adamw_flat_dev": [c_void_p] * 6 + [c_float, c_void_p, c_void_p, c_int32, c_int64, c_void_p],
    "vr_adamw_flat_dev_capped": [c_void_p] * 6 + [c_float, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p],
    "vr_cast_transpose_batch": [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p],
    "vr_ln_fwd": [c_void_p] * 7 + [c_int32, c_int32, c_int32, c_float, c_int32, c_void_p],
    "vr_ln_bwd": [c_void_p] * 13 + [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "vr_ln_grad_reduce": [ctypes.POINTER(LnGradSlot), c_int32, c_int32, c_void_p],
    "vr_attn_fwd": [c_void_p] * 4 + [c_int32] * 4 + [c_float, c_int32, c_void_p],
    "vr_attn_bwd": [c_void_p] * 7 + [c_int32] * 4 + [c_float, c_int32, c_void_p],
    "vr_softce": [c_void_p] * 4 + [c_int32, c_int32, c_float, c_void_p],
    "vr_softce_train": [c_void_p] * 3 + [c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_void_p],
    "vr_colsum": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, RowMap, c_void_p],
    "vr_scale_mask_cast": [c_void_p] * 4 + [c_int32] * 4 + [c_void_p],
    "vr_conv3x3_wgrad": [c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p],
    "vr_conv3x3": [c_void_p, c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p],
    "vr_conv3x3_bias_relu": [c_void_p] * 5 + [c_int32] * 6 + [c_void_p],
    "vr_conv3x3_res": [c_void_p] * 4 + [c_int32] * 6 + [c_void_p],
    "vr_conv3x3_bias_relu_patch": [c_void_p] * 5 + [c_int32] * 7 + [c_void_p],
    "vr_conv_w_flip": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p],
    "vr_conv1_direct": [c_void_p] * 4 + [c_int32] * 6 + [c_void_p],
    "vr_token_mix": [c_void_p] * 6 + [c_int32] * 11 + [c_float] * 6 + [c_void_p],
    "vr_token_mean": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "vr_token_mean_bwd": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "vr_batchsum": [c_void_p, c_void_p, c_int32, c_int64, c_void_p],
    "vr_im2col_patch": [c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p],
    "vr_im2col_patch_map": [c_void_p, c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p],
    "vr_embed_cls": [c_void_p] * 4 + [c_int32] * 4 + [c_void_p],
    "vr_sr_im2col": [c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p],
    "vr_sr_col2im": [c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p],
    "vr_sr_resid": [c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p],
    "vr_sr_resid_bwd": [c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p],
    "vr_mask_rows": [c_void_p, c_void_p] + [c_int32] * 3 + [c_void_p],
    "vr_zero_ranges": [c_void_p, ctypes.POINTER(ZeroRanges), c_void_p],
    "vr_relayout": [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int64, c_int32, c_int32, c_void_p],
    "vr_im2col3x3": [c_void_p, c_void_p] + [c_int32] * 8 + [c_void_p],
    "vr_col2im3x3": [c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p],
    "vr_bn_stats": [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p],
    "vr_bn_finalize": [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_float] + [c_void_p] * 7 + [c_int32, c_void_p],
    "vr_bn_relu": [c_void_p] * 5 + [c_int64, c_int32, c_int32, c_int32, c_void_p],
    "vr_bn_bwd": [c_void_p] * 9 + [c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p],
    "vr_patch_unfold": [c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p],
    "vr_bn_relu_patch": [c_void_p] * 5 + [c_int32] * 7 + [c_void_p],
    "vr_bn_bwd_patch": [c_void_p] * 9 + [c_int32] * 8 + [c_void_p],
    "vr_conv3x3_res_patch": [c_void_p] * 4 + [c_int32] * 7 + [c_void_p],
}

_lib = None


def lib():
    """Load (once) and return the library; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libvitres_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C vit-search_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise RuntimeError("libvitres_hip.so does not export %s" % name) from e
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = L
    return _lib


_ERR = {-1: "VR_EINVAL (bad argument)", -2: "VR_EALIGN (pointer / leading-dimension alignment)",
        -3: "VR_EUNSUPPORTED (shape or dtype not supported)"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "hipError_t %d" % rc)))
