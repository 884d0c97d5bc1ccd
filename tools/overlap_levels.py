"""Where two forwards in flight lose time: per-stage HIP-event times (swf_model_forward_profiled) of eager forwards issued from one
host thread per stream, 1 thread against 2 (and 3).  ctypes releases the GIL inside the call, so the threads' launch chains overlap
on the GPU the way the runner's lanes do (eager instead of hipGraph).  Prints one JSON line.

    python tools/overlap_levels.py [--threads 1 2]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    entry.build()
    from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
    from swin_unet_image_fusion_amd import _lib as L
    from swin_unet_image_fusion_amd.modules import _ptr
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    cfg = CONFIGS["win8"]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=0)
    model.to(dev)
    b, n = args.batch, 256
    ir, vis = (torch.from_numpy(a).to(dev) for a in synthetic_pair(b, n, n, seed_ir=1, seed_vis=2))
    model(ir, vis)
    lib, desc = L.lib(), model._model_desc()
    arena = model._get_arena(dev)
    packed = model._get_packed(arena)
    need = lib.swf_model_workspace_bytes(C.byref(desc), b, n, n)
    nlev = len(model.in_dims_list)
    nseg = 4 * nlev + 1
    names = []
    for s in range(nlev):
        names += [f"enc{s}_patch", f"enc{s}_blocks"]
    for j in range(nlev):
        names += [f"dec{nlev - 1 - j}_blocks", f"dec{nlev - 1 - j}_patch"]
    names.append("head")
    torch.cuda.synchronize()
    result = []
    for nt in args.threads:
        streams = [torch.cuda.Stream(device=dev) for _ in range(nt)]
        wss = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(nt)]
        outs = [torch.empty((b, 1, n, n), dtype=torch.float32, device=dev) for _ in range(nt)]
        acc = [[0.0] * nseg for _ in range(nt)]
        start = threading.Barrier(nt)
        walls = [0.0] * nt

        def work(i):
            seg = (C.c_float * nseg)()
            torch.cuda.set_device(dev)
            for it in range(args.iters + 4):
                if it == 4:
                    start.wait()
                    t0 = time.perf_counter()
                L.check(lib.swf_model_forward_profiled(C.byref(desc), _ptr(arena), packed.data_ptr(), _ptr(ir), _ptr(vis), _ptr(outs[i]), b, n, n,
                                                       wss[i].data_ptr(), need, seg, nseg, streams[i].cuda_stream))
                if it >= 4:
                    for k in range(nseg):
                        acc[i][k] += float(seg[k])
            walls[i] = time.perf_counter() - t0

        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        seg_ms = [sum(acc[i][k] for i in range(nt)) / nt / args.iters for k in range(nseg)]
        wall = max(walls)
        result.append({"threads": nt, "pairs_per_s": round(nt * args.iters * b / wall, 1), "ms_per_forward_in_its_stream": round(sum(seg_ms), 4),
                       "segments_us": {nm: round(v * 1e3, 1) for nm, v in zip(names, seg_ms)}})
    print(json.dumps({"what": f"B={b} 256x256 win8: eager profiled forwards from N host threads, one stream each", "runs": result}))


if __name__ == "__main__":
    main()
