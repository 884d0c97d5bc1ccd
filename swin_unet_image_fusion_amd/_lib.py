"""ctypes binding of libswinfuse.so (include/swinfuse.h).

No torch types cross this boundary: tensors are handed over as `data_ptr()` integers and the
current HIP stream as a `void*`.  The library is built in-tree by `__graft_entry__.build()`
(hipcc --offload-arch=gfx950); if it is missing, importing anything that computes fails loudly —
there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWF_LIB_PATH") or os.path.join(_HERE, "libswinfuse.so")   # SWF_LIB_PATH: A/B builds of the library

SWF_MAX_LEVELS = 8
PREC_FP32, PREC_FAST = 0, 1

# status codes (swf_status)
OK, ERR_NULL, ERR_BAD_SHAPE, ERR_PAD, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_HIP = 0, -1, -2, -3, -4, -5, -6

c_float_p = C.c_void_p  # device pointers travel as integers


class Linear(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class Norm(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
                ("win_h", C.c_int32), ("win_w", C.c_int32), ("shift", C.c_int32)]


class AttnParams(C.Structure):
    _fields_ = [("q", Linear), ("k", Linear), ("v", Linear), ("proj", Linear), ("bias_table", C.c_void_p)]


class BlockDesc(C.Structure):
    _fields_ = [("attn", AttnDesc), ("hidden", C.c_int32), ("cross", C.c_int32), ("precision", C.c_int32), ("schedule", C.c_int32)]


class BlockStreamParams(C.Structure):
    _fields_ = [("ln1", Norm), ("attn", AttnParams), ("ln2", Norm), ("fc1", Linear), ("fc2", Linear)]


class PatchParams(C.Structure):
    _fields_ = [("conv", Linear), ("ln", Norm)]


class HeadParams(C.Structure):
    _fields_ = [("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p)]


class HeadGrads(C.Structure):
    _fields_ = [("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p)]


class ModelDesc(C.Structure):
    _fields_ = [("levels", C.c_int32), ("in_dims", C.c_int32 * SWF_MAX_LEVELS), ("out_dims", C.c_int32 * SWF_MAX_LEVELS),
                ("heads", C.c_int32), ("head_dim", C.c_int32 * SWF_MAX_LEVELS), ("mlp_ratio", C.c_int32),
                ("win_h", C.c_int32), ("win_w", C.c_int32), ("merge_h", C.c_int32), ("merge_w", C.c_int32),
                ("head_ksize", C.c_int32), ("precision", C.c_int32), ("schedule", C.c_int32)]


P = C.POINTER
_i32, _i64, _sz, _vp = C.c_int32, C.c_int64, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); this table is also what tests check against include/swinfuse.h
SIGNATURES = {
    "swf_version": (C.c_int, []),
    "swf_last_error_string": (C.c_char_p, []),
    "swf_status_string": (C.c_char_p, [C.c_int]),
    "swf_window_attention_fwd": (C.c_int, [P(AttnDesc), P(AttnParams), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_window_attention_fwd_prec": (C.c_int, [P(AttnDesc), _i32, P(AttnParams), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_window_attention_workspace_bytes": (_sz, [P(AttnDesc), _i32, _i32, _i32]),
    "swf_attn_halfblock_fwd": (C.c_int, [P(BlockDesc), P(BlockStreamParams), P(BlockStreamParams), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_mlp_halfblock_fwd": (C.c_int, [P(BlockDesc), P(BlockStreamParams), P(BlockStreamParams), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_basic_block_fwd": (C.c_int, [P(BlockDesc), P(BlockStreamParams), P(BlockStreamParams), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_basic_block_workspace_bytes": (_sz, [P(BlockDesc), _i32, _i32, _i32]),
    "swf_basic_block_packed_bytes": (_sz, [P(BlockDesc)]),
    "swf_basic_block_pack": (C.c_int, [P(BlockDesc), P(BlockStreamParams), P(BlockStreamParams), _vp, _sz, _vp]),
    "swf_basic_block_fwd_packed": (C.c_int, [P(BlockDesc), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "swf_basic_block_bwd_workspace_bytes": (_sz, [P(BlockDesc), _i32, _i32, _i32]),
    "swf_basic_block_bwd": (C.c_int, [P(BlockDesc), P(BlockStreamParams), P(BlockStreamParams), _vp, _vp, _vp, _vp, _vp, _vp,
                                      P(BlockStreamParams), P(BlockStreamParams), _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_window_attention_bwd_workspace_bytes": (_sz, [P(AttnDesc), _i32, _i32, _i32]),
    "swf_window_attention_bwd": (C.c_int, [P(AttnDesc), P(AttnParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, P(AttnParams), _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_mlp_bwd_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "swf_mlp_bwd": (C.c_int, [P(Linear), P(Linear), _vp, _vp, _vp, P(Linear), P(Linear), _i64, _i32, _i32, _vp, _sz, _vp]),
    "swf_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i32]),
    "swf_layernorm_bwd": (C.c_int, [P(Norm), _vp, _vp, _vp, P(Norm), _i64, _i32, _vp, _sz, _vp]),
    "swf_patch_layer_bwd_workspace_bytes": (_sz, [_i32] * 8),
    "swf_patch_layer_bwd": (C.c_int, [P(PatchParams), _vp, _vp, _vp, P(PatchParams), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_reflect_pad_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "swf_final_head_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "swf_final_head_bwd": (C.c_int, [P(HeadParams), _vp, _vp, _vp, _vp, _vp, P(HeadGrads), _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_final_head_batch_stats": (C.c_int, [P(HeadParams), _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_add_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "swf_block_pair4_fwd": (C.c_int, [P(BlockDesc), P(BlockStreamParams), P(BlockStreamParams), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_patch_merge_fwd": (C.c_int, [P(PatchParams), _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_merge_out_shape": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, P(_i32), P(_i32), P(_i32), P(_i32)]),
    "swf_patch_unmerge_fwd": (C.c_int, [P(PatchParams), _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_patch_workspace_bytes": (_sz, [_i32] * 10),
    "swf_final_head_fwd": (C.c_int, [P(HeadParams), _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_linear_fwd": (C.c_int, [P(Linear), _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "swf_mlp_workspace_bytes": (_sz, [_i32, _i64, _i32, _i32]),
    "swf_mlp_fwd": (C.c_int, [_i32, P(BlockStreamParams), P(BlockStreamParams), _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _sz, _vp]),
    "swf_linear_workspace_bytes": (_sz, [_i32, _i64, _i32, _i32]),
    "swf_linear_fwd_prec": (C.c_int, [P(Linear), _i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_layernorm_fwd": (C.c_int, [P(Norm), _vp, _vp, _i64, _i32, _i32, _vp]),
    "swf_reflect_pad_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "swf_crop_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "swf_nchw_to_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "swf_nhwc_to_nchw": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "swf_bgr8_to_ycrcb_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "swf_gray8_to_unit_fwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "swf_ycrcb_to_rgb_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "swf_model_param_count": (_i32, [P(ModelDesc)]),
    "swf_model_param_info": (C.c_int, [P(ModelDesc), _i32, C.c_char_p, _sz, P(_i64), P(_i64)]),
    "swf_model_arena_elems": (_i64, [P(ModelDesc)]),
    "swf_model_workspace_bytes": (_sz, [P(ModelDesc), _i32, _i32, _i32]),
    "swf_model_forward": (C.c_int, [P(ModelDesc), _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_model_packed_bytes": (_sz, [P(ModelDesc)]),
    "swf_model_pack_weights": (C.c_int, [P(ModelDesc), _vp, _vp, _sz, _vp]),
    "swf_model_forward_packed": (C.c_int, [P(ModelDesc), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "swf_model_forward_checked": (C.c_int, [P(ModelDesc), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp, _vp]),
    "swf_model_forward_profiled": (C.c_int, [P(ModelDesc), _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, P(C.c_float), _i32, _vp]),
    "swf_tensors_equal": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


class SwinFuseLibraryMissing(ImportError):
    pass


def lib() -> C.CDLL:
    """Load libswinfuse.so (once).  Raises — never falls back — when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwinFuseLibraryMissing(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status: int) -> None:
    """Map swf_status to the exception classes the reference raises:
    bad shapes -> ValueError (einops / mode errors), reflect pad >= dim -> RuntimeError
    (torch, a006:128), everything else -> RuntimeError."""
    if status == OK:
        return
    msg = lib().swf_last_error_string().decode(errors="replace")
    kind = lib().swf_status_string(status).decode()
    text = f"libswinfuse: {kind}: {msg}"
    if status == ERR_BAD_SHAPE:
        raise ValueError(text)
    if status == ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    raise RuntimeError(text)
