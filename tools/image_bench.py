"""End-to-end time of the reference's inference step for in-memory images (a017_test.py:55-90): uint8 IR + BGR visible image ->
BGR2YCrCb split -> [0,1] -> MyModel.forward -> clamp -> YCrCb -> RGB uint8, every step a kernel of libswinfuse on one stream
(imaging.fuse_images).  Prints one JSON line per shape (profiles/r03_image_level.json).  The colour steps are checked against a numpy
restatement of OpenCV's formulas only (cv2 is not installed here): parity unpinned for them.

    python tools/image_bench.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import __graft_entry__ as entry


def main():
    entry.build()
    from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into
    from swin_unet_image_fusion_amd.imaging import fuse_images
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    cfg = CONFIGS["win8"]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=0)
    model.to(dev)
    g = torch.Generator(device="cpu").manual_seed(3)
    out = []
    for b, h, w in ((1, 256, 256), (1, 512, 640), (16, 256, 256)):
        ir = torch.randint(0, 256, (b, h, w), dtype=torch.uint8, generator=g).to(dev)
        vis = torch.randint(0, 256, (b, h, w, 3), dtype=torch.uint8, generator=g).to(dev)
        for _ in range(3):
            rgb = fuse_images(model, ir, vis)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            rgb = fuse_images(model, ir, vis)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        # the model alone on the same inputs
        from swin_unet_image_fusion_amd.imaging import prepare_pair
        irf, vy, crcb = prepare_pair(ir, vis)
        for _ in range(2):
            model(irf, vy)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            model(irf, vy)
        e1.record()
        torch.cuda.synchronize()
        ms_model = e0.elapsed_time(e1) / iters
        assert rgb.shape == (b, h, w, 3) and rgb.dtype == torch.uint8
        out.append({"shape": f"B={b} {h}x{w} uint8 IR + BGR visible", "ms_end_to_end": round(ms, 4), "ms_model_forward": round(ms_model, 4),
                    "pairs_per_s": round(b / ms * 1e3, 1), "eager": True})
    print(json.dumps({"what": "imaging.fuse_images (a017:55-90 for in-memory images), eager launches, HIP events", "colour_parity": "unpinned (cv2 absent)", "runs": out}))


if __name__ == "__main__":
    main()
