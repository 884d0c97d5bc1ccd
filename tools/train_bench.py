"""Time of one training step of the module-by-module autograd path: forward under autograd + backward of every
parameter (exact fp32, atomics-free) + a plain SGD update, model.train() as the reference trains (a016:137).  Prints one JSON line.
The loss is a smooth stand-in (mean squared distance to max(ir, vis)): the reference's loss needs kornia, absent here.

    python tools/train_bench.py [--batch 4] [--size 128] [--config win8] [--iters 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--config", default="win8")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    entry.build()
    from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
    dev = torch.device("cuda:0")
    cfg = CONFIGS[args.config]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True)))
    load_recipe_into(model, seed=0, flavor="kaiming")
    model.to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    ir, vis = (torch.from_numpy(a).to(dev) for a in synthetic_pair(args.batch, args.size, args.size, seed_ir=1, seed_vis=2))
    tgt = torch.maximum(ir, vis)
    times = {"forward": 0.0, "backward": 0.0, "update": 0.0}
    losses = []
    for it in range(args.iters + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(ir, vis)
        loss = (out - tgt).square().mean()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step(); model.refresh_weights()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        losses.append(float(loss))
        if it:
            times["forward"] += t1 - t0; times["backward"] += t2 - t1; times["update"] += t3 - t2
    ms = {k: round(v / args.iters * 1e3, 2) for k, v in times.items()}
    total = sum(ms.values())
    print(json.dumps({"what": f"training step B={args.batch} {args.size}x{args.size} {args.config}, model.train(), autograd path, SGD",
                      "ms": ms, "ms_per_step": round(total, 2), "pairs_per_s": round(args.batch / total * 1e3, 1),
                      "loss_first_last": [losses[0], losses[-1]]}))


if __name__ == "__main__":
    main()
