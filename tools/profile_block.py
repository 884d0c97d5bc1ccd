"""Launch one BasicBlock unit repeatedly (for rocprofv3 / PMC runs and A/B timing in one process).

    python tools/profile_block.py --level 0 --iters 20 [--precision fast|fp32] [--shift 1] [--cross 1] [--schedule throughput]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=0)
    ap.add_argument("--decoder", type=int, default=0)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--precision", default="fast")
    ap.add_argument("--shift", type=int, default=1)
    ap.add_argument("--cross", type=int, default=1)
    ap.add_argument("--schedule", default="latency", choices=["latency", "throughput"])
    a = ap.parse_args()
    entry.build()
    from swin_unet_image_fusion_amd import CONFIGS, MyModel, _lib as L, load_recipe_into
    from swin_unet_image_fusion_amd.modules import _ptr, _stream, _workspace
    torch.set_grad_enabled(False)
    cfg = CONFIGS["win8"]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=0)
    model.to("cuda:0")
    stage = model.decoder_list[len(cfg.in_dims_list) - 1 - a.level][0] if a.decoder else model.encoder_list[a.level][3]
    grp = stage.cross_att_block if a.cross else stage.self_att_block
    blk = grp.shifted_window_block if a.shift else grp.normal_window_block
    c = blk.in_out_dims
    h = w = a.size >> (a.level + 1)
    x = torch.randn(a.batch, h, w, c, device="cuda:0")
    y = torch.randn(a.batch, h, w, c, device="cuda:0")
    ox, oy = torch.empty_like(x), torch.empty_like(y)
    lib = L.lib()
    desc = blk._desc(a.precision)
    desc.schedule = 1 if a.schedule == "throughput" else 0
    px, py = blk._stream_params("x"), blk._stream_params("y")
    ws, wsn = _workspace(lib.swf_basic_block_workspace_bytes(C.byref(desc), a.batch, h, w), x.device)
    st = _stream(x.device)

    def run():
        L.check(lib.swf_basic_block_fwd(C.byref(desc), C.byref(px), C.byref(py), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy),
                                        a.batch, h, w, ws, wsn, st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    if os.environ.get("SWF_CHECK_DET"):
        ref_x, ref_y = ox.clone(), oy.clone()
        bad = 0
        for it in range(20):
            ox.zero_(); oy.zero_()
            run()
            torch.cuda.synchronize()
            if not (torch.equal(ox, ref_x) and torch.equal(oy, ref_y)):
                bad += 1
                dx = (ox != ref_x).nonzero()
                dy = (oy != ref_y).nonzero()
                if bad <= 3:
                    print("  mismatch iter", it, "x:", dx.shape[0], dx[:3].tolist(), "y:", dy.shape[0], dy[:3].tolist(),
                          float((ox - ref_x).abs().max()), float((oy - ref_y).abs().max()))
        print(f"determinism: {bad}/20 runs differ (shift={a.shift} cross={a.cross} dec={a.decoder})")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"level {a.level} dec={a.decoder} C={c} hid={blk.mlp_hidden_dims} map {h}x{w} B={a.batch} prec={a.precision}: "
          f"{e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per block launch")


if __name__ == "__main__":
    main()
