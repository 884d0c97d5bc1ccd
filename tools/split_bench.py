"""Experiment: one forward of B pairs as k concurrent sub-batch chains (k hipGraphs replayed on k streams) against the single chain.
The forward is a chain of dependent launches whose deep levels are latency-bound; sub-batches on separate streams let one
chain's latency-bound launches run under another chain's throughput-bound ones.  Prints one JSON line.

    python tools/split_bench.py [--batch 16] [--size 256] [--iters 50]
"""
import argparse
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
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--config", default="win8")
    ap.add_argument("--chains", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--in-flight", type=int, nargs="*", default=[1, 2, 3])
    args = ap.parse_args()
    entry.build()
    from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
    from swin_unet_image_fusion_amd.shard import ShardedFusion
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    cfg = CONFIGS[args.config]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=0)
    model.to(dev)
    b, n = args.batch, args.size
    ir, vis = (torch.from_numpy(a).to(dev) for a in synthetic_pair(b, n, n, seed_ir=1, seed_vis=2))
    ref = model(ir, vis).clone()
    runs = []
    main_s = torch.cuda.current_stream(dev)
    for k in args.chains:
        if b % k:
            continue
        per = b // k
        runners = [ShardedFusion(model, use_graph=True) for _ in range(k)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        outs = [r.local_forward(ir[i * per:(i + 1) * per].contiguous(), vis[i * per:(i + 1) * per].contiguous()) for i, r in enumerate(runners)]
        torch.cuda.synchronize()
        err = float((torch.cat(outs) - ref).abs().max())

        def step():
            for s, r in zip(streams, runners):
                s.wait_stream(main_s)
                with torch.cuda.stream(s):
                    r._graph.replay()
            for s in streams:
                main_s.wait_stream(s)

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        runs.append({"chains": k, "pairs_per_chain": per, "ms_per_forward": round(ms, 4), "pairs_per_s": round(b / ms * 1e3, 1),
                     "max_abs_diff_vs_one_chain": err})
        del runners, streams, outs
    # steps in flight: k runners of the FULL batch on k streams, consecutive steps alternate between them and only the end is joined
    piped = []
    for k in args.in_flight:
        runners = [ShardedFusion(model, use_graph=True) for _ in range(k)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        for r in runners:
            r.local_forward(ir, vis)
        torch.cuda.synchronize()
        err = max(float((r._static[2] - ref).abs().max()) for r in runners)

        def run(nsteps):
            for s in streams:
                s.wait_stream(main_s)
            for i in range(nsteps):
                with torch.cuda.stream(streams[i % k]):
                    runners[i % k]._graph.replay()
            for s in streams:
                main_s.wait_stream(s)

        run(6)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(args.iters)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        piped.append({"steps_in_flight": k, "ms_per_step": round(ms, 4), "pairs_per_s": round(b / ms * 1e3, 1), "max_abs_diff": err})
        del runners, streams
    print(json.dumps({"what": f"B={b} {n}x{n} {args.config}: k sub-batch graphs replayed on k streams per forward", "runs": runs,
                      "steps_in_flight": piped}))


if __name__ == "__main__":
    main()
