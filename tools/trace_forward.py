"""Per-kernel timeline of ONE eager forward (no hipGraph) from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace -d gpurun_out/tr -o t -- python3 tools/trace_forward.py run
    python3 tools/trace_forward.py report gpurun_out/tr/t_results.db   # prints the last forward, kernel by kernel
"""
import sqlite3
import sys


def run():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from torch import nn
    import __graft_entry__ as entry
    entry.build()
    from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
    torch.set_grad_enabled(False)
    cfg = CONFIGS[os.environ.get("TRACE_CONFIG", "win8")]   # TRACE_CONFIG / TRACE_BATCH / TRACE_SIZE: other BASELINE configs
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=0)
    model.to("cuda:0")
    model.precision = "fast"
    model.schedule = os.environ.get("TRACE_SCHEDULE", "latency")   # "throughput": the kernels ShardedFusion's lanes replay
    size = int(os.environ.get("TRACE_SIZE", "256"))
    ir, vis = synthetic_pair(int(os.environ.get("TRACE_BATCH", "16")), size, size)
    ir, vis = torch.from_numpy(ir).cuda(), torch.from_numpy(vis).cuda()
    for _ in range(4):
        model(ir, vis)
    torch.cuda.synchronize()


def report(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                            f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    # last forward = everything after the last but one head kernel (the last launch of a forward)
    idx = [i for i, r in enumerate(rows) if "head_conv2" in r[0] or "head_fused3" in r[0]]
    lo = idx[-2] + 1 if len(idx) >= 2 else 0
    fw = rows[lo:idx[-1] + 1]
    t0 = fw[0][1]
    busy = 0
    prev_end = None
    gaps = 0
    for name, s, e, gx, gy, gz, wx in fw:
        short = name.split("(")[0].replace("void ", "").replace("swf::", "").replace(".kd", "")[:60]
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        gaps += max(gap, 0)
        busy += e - s
        print(f"{(s - t0) / 1e3:9.1f} us  +{gap:5.1f}  {(e - s) / 1e3:7.1f} us  grid {gx // max(wx,1):6d}x{gy}x{gz}  {short}")
        prev_end = e
    print(f"kernels {len(fw)}  busy {busy / 1e3:.1f} us  gaps {gaps:.1f} us  span {(fw[-1][2] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
