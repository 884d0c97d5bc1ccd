"""Stand-alone module entries at the level-0 shape (B=16 128x128 map, C=24, 8 heads x 3, 8x8 windows, shifted, hidden 96), timed with
events on the launch stream through the C-ABI (NHWC tensors, no layout changes): WindowAttention.forward (a001:448-474),
AddAndLayerNormWithOtherModule around AutoPathWinAtt / AutoPathMLP (a004), AutoPathMLP.forward (a003) — fast tier (one launch of the
block kernel with the other half compiled out) and exact tier.  Prints one JSON line (profiles/r03_a1_standalone.json).

    python tools/standalone_bench.py [--batch 16] [--size 128] [--iters 20]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--channels", type=int, default=24, help="24 / 48 / 96 (levels 0-2; hidden 4 x channels)")
    a = ap.parse_args()
    entry.build()
    from swin_unet_image_fusion_amd import BasicBlock, _lib as L, load_recipe_into
    from swin_unet_image_fusion_amd.modules import _ptr, _stream, _workspace
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    c, hid = a.channels, 4 * a.channels
    blk = BasicBlock(c, 8, c // 8, (8, 8), True, True, True, True, 0.0, 0.0, hid, nn.ELU(inplace=True), 0.0).eval()
    load_recipe_into(blk, seed=0)
    blk.to(dev)
    b, h, w = a.batch, a.size, a.size
    n = b * h * w
    x, y = torch.randn(b, h, w, c, device=dev), torch.randn(b, h, w, c, device=dev)
    ox, oy = torch.empty_like(x), torch.empty_like(y)
    lib, st = L.lib(), _stream(dev)
    wa = blk.auto_path_win_att.window_attention_x
    adesc, aprm = wa._desc(), wa._params()
    px, py = blk._stream_params("x"), blk._stream_params("y")
    tbl = 15 * 15

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3   # us

    out = {"shape": f"B={b} {h}x{w} map, C={c}, 8 heads x {c // 8}, 8x8 windows (shifted), hidden {hid}; NHWC fp32 in HBM", "unit": "us per call", "entries": {}}
    for prec_name, prec in (("fast", L.PREC_FAST), ("fp32", L.PREC_FP32)):
        bdesc = blk._desc(prec_name)
        ws, wsn = _workspace(max(lib.swf_window_attention_workspace_bytes(C.byref(adesc), b, h, w),
                                 lib.swf_basic_block_workspace_bytes(C.byref(bdesc), b, h, w),
                                 lib.swf_mlp_workspace_bytes(prec, n, c, hid)), dev)
        calls = {
            # cross form: q from one stream, k = v from the other (a002:67-82)
            "WindowAttention.forward (swf_window_attention_fwd_prec)": (
                lambda: L.check(lib.swf_window_attention_fwd_prec(C.byref(adesc), prec, C.byref(aprm), _ptr(x), _ptr(y), _ptr(y), None, _ptr(ox), b, h, w, ws, wsn, st)),
                3 * n * c * 4 + (4 * c * c + 4 * c + tbl) * 4),
            "AddAndLayerNorm(AutoPathWinAtt) both streams (swf_attn_halfblock_fwd)": (
                lambda: L.check(lib.swf_attn_halfblock_fwd(C.byref(bdesc), C.byref(px), C.byref(py), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy), b, h, w, ws, wsn, st)),
                4 * n * c * 4 + 2 * (4 * c * c + 6 * c + tbl) * 4),
            "AddAndLayerNorm(AutoPathMLP) both streams (swf_mlp_halfblock_fwd)": (
                lambda: L.check(lib.swf_mlp_halfblock_fwd(C.byref(bdesc), C.byref(px), C.byref(py), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy), b, h, w, ws, wsn, st)),
                4 * n * c * 4 + 2 * (2 * c * hid + hid + 3 * c) * 4),
            "AutoPathMLP.forward both streams (swf_mlp_fwd)": (
                lambda: L.check(lib.swf_mlp_fwd(prec, C.byref(px), C.byref(py), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy), n, c, hid, ws, wsn, st)),
                4 * n * c * 4 + 2 * (2 * c * hid + hid + c) * 4),
        }
        for name, (fn, alg_bytes) in calls.items():
            us = timed(fn)
            out["entries"].setdefault(name, {})[prec_name] = {"us": round(us, 1), "algorithmic_bytes": alg_bytes,
                                                               "frac_hbm": round(alg_bytes / (us * 1e-6) / 8e12, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
