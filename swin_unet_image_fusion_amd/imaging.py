"""Image-space wrapper around the model: the reference's inference loop (a017_test.py:55-90) for in-memory images.

`fuse_images(model, ir_u8, vis_bgr_u8)` = split the visible image into Y / CrCb (cv2 BGR2YCrCb on uint8, a015:89),
scale to [0,1], run `model(ir, vis_y)`, clamp, re-attach CrCb, convert back to RGB — every step a HIP kernel of
libswinfuse on the current stream, no host round trip.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import _lib as L
from .modules import _stream


def _u8(t: Tensor, what: str) -> Tensor:
    if t.dtype != torch.uint8 or not t.is_cuda:
        raise RuntimeError(f"{what}: expected a uint8 tensor on the GPU, got {t.dtype} on {t.device}")
    return t.contiguous()


def prepare_pair(ir_u8: Tensor, vis_bgr_u8: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """ir_u8 (B,H,W) gray, vis_bgr_u8 (B,H,W,3) BGR as cv2.imread returns them -> ir (B,1,H,W), vis_y (B,1,H,W),
    crcb (B,2,H,W), float32 in [0,1]."""
    ir_u8, vis_bgr_u8 = _u8(ir_u8, "ir"), _u8(vis_bgr_u8, "vis")
    if vis_bgr_u8.dim() != 4 or vis_bgr_u8.shape[-1] != 3 or ir_u8.shape != vis_bgr_u8.shape[:3]:
        raise ValueError(f"expected ir (B,H,W) and vis (B,H,W,3), got {tuple(ir_u8.shape)} and {tuple(vis_bgr_u8.shape)}")
    b, h, w = ir_u8.shape
    dev = ir_u8.device
    ir = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
    vis_y = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
    crcb = torch.empty((b, 2, h, w), dtype=torch.float32, device=dev)
    lib, st = L.lib(), _stream(dev)
    L.check(lib.swf_gray8_to_unit_fwd(ir_u8.data_ptr(), ir.data_ptr(), b * h * w, st))
    L.check(lib.swf_bgr8_to_ycrcb_fwd(vis_bgr_u8.data_ptr(), vis_y.data_ptr(), crcb.data_ptr(), b, h, w, st))
    return ir, vis_y, crcb


def finish(fused_y: Tensor, crcb: Tensor, as_uint8: bool = False) -> Tensor:
    """fused_y (B,1,H,W) unclamped + crcb (B,2,H,W) -> RGB: float32 (B,3,H,W), or uint8 (B,H,W,3) quantised like
    torchvision.utils.save_image (a017:90)."""
    b, _, h, w = fused_y.shape
    dev = fused_y.device
    fused_y, crcb = fused_y.contiguous(), crcb.contiguous()
    lib, st = L.lib(), _stream(dev)
    if as_uint8:
        out = torch.empty((b, h, w, 3), dtype=torch.uint8, device=dev)
        L.check(lib.swf_ycrcb_to_rgb_fwd(fused_y.data_ptr(), crcb.data_ptr(), None, out.data_ptr(), b, h, w, st))
    else:
        out = torch.empty((b, 3, h, w), dtype=torch.float32, device=dev)
        L.check(lib.swf_ycrcb_to_rgb_fwd(fused_y.data_ptr(), crcb.data_ptr(), out.data_ptr(), None, b, h, w, st))
    return out


def fuse_images(model, ir_u8: Tensor, vis_bgr_u8: Tensor, as_uint8: bool = True) -> Tensor:
    ir, vis_y, crcb = prepare_pair(ir_u8, vis_bgr_u8)
    with torch.no_grad():
        return finish(model(ir, vis_y), crcb, as_uint8=as_uint8)
