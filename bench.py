"""bench.py — fused image-pairs/s of the Swin-UNet fusion forward path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one MyModel.forward (the whole hot path, through the C-ABI) over one batch of synthetic
IR/visible pairs already resident in HBM, plus — for N > 1 — the RCCL all-gather of the fused output
(SURVEY.md §8e).  Weak scaling: every GPU processes `--batch` pairs (BASELINE.json config 4 = 16 x 8).
Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launch stream around
the dominant unit (the level-0 window-attention BasicBlock launch); `cpu_baseline` times the CPU oracle
(a port of the reference's PyTorch-CPU path) on a bounded sample on rank 0 at N=1.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch import nn  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
BYTES_PER_ELEM = 4          # the residual stream is stored fp32 (DESIGN.md "Data layout")
TRAFFIC_FILE = "r03_traffic.json"    # committed PMC summary of the level-0 block kernel (tools/pmc_levels.sh, tools/traffic_summary.py)
PMC_LEVELS_FILE = "r03_pmc_levels.json"   # committed per-level counter summary (tools/pmc_levels.sh, tools/pmc_levels_summary.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="pairs per GPU (BASELINE configs 2/4)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--height", type=int, default=0, help="non-square inputs: height (default: --size)")
    ap.add_argument("--width", type=int, default=0, help="non-square inputs: width (default: --size)")
    ap.add_argument("--config", default="win8")
    ap.add_argument("--precision", default="fast", choices=["fast", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="steps kept in flight per GPU (hipGraph lanes on streams measured to sit on distinct hardware queues; the runner "
                         "keeps fewer if the device has fewer free queues; 1 = one chain)")
    ap.add_argument("--force-collective", action="store_true",
                    help="one GPU only: issue the per-step all-gather on a one-rank RCCL group (rehearses the N>1 stream / queue pattern)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=4, help="pairs per CPU-baseline forward")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the control flow (process group, barriers, pipelined step_async, MAX all-reduce of the elapsed time, rank-0 "
                         "JSON) on CPU tensors over gloo with a stub forward: no GPU, no library, no number worth reading (tests/test_bench_dryrun.py)")
    ap.add_argument("--no-levels", action="store_true", help="skip the per-level roofline pass (swf_model_forward_profiled)")
    return ap.parse_args()


def level0_block_roofline(model, batch, size, precision, iters=20, height=0, width=0):
    """Time the dominant unit — the level-0 BasicBlock (window attention + MLP on the full-resolution
    C=out_dims[0] map, both streams) — alone, with HIP events on the launch stream, and price it against
    HBM.  Algorithmic bytes per launch (attention half-block, both streams, e=4):
    4*N*C*e + weights.  The fused kernel reads and writes each stream once, so the MLP half adds none."""
    from swin_unet_image_fusion_amd import _lib as L
    from swin_unet_image_fusion_amd.modules import _precision_code, _ptr, _stream, _workspace

    blk = model.encoder_list[0][3].self_att_block.shifted_window_block
    c = blk.in_out_dims
    mh, mw = model.merging_size
    # the level-0 map as the model runs it: merged (reflect-padded to the merge size), then padded to a multiple of the window
    wh, ww = blk.window_size
    h, w = -(-(-(-(height or size) // mh)) // wh) * wh, -(-(-(-(width or size) // mw)) // ww) * ww
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(batch, h, w, c, generator=g).to(dev)
    y = torch.randn(batch, h, w, c, generator=g).to(dev)
    ox, oy = torch.empty_like(x), torch.empty_like(y)
    lib = L.lib()
    desc = blk._desc(precision)
    px, py = blk._stream_params("x"), blk._stream_params("y")
    stream = _stream(dev)
    nbytes = lib.swf_basic_block_packed_bytes(C.byref(desc))
    if nbytes:      # fast tier: weights pre-packed once (as the model path does), each launch = exactly the fused kernel
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(lib.swf_basic_block_pack(C.byref(desc), C.byref(px), C.byref(py), packed.data_ptr(), nbytes, stream))
        kname = (f"fused window-attention BasicBlock kernel, C={c} hidden={blk.mlp_hidden_dims} win={blk.window_size[0]} "
                 "(one launch = level-0 shifted-window BasicBlock, both streams; kernel symbol in profiles/)")

        def run():
            L.check(lib.swf_basic_block_fwd_packed(C.byref(desc), packed.data_ptr(), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy),
                                                   batch, h, w, stream))
    else:           # exact tier: the unit is a sequence of kernels
        ws, wsn = _workspace(lib.swf_basic_block_workspace_bytes(C.byref(desc), batch, h, w), dev)
        kname = "level-0 shifted-window BasicBlock unit (swf_basic_block_fwd, exact tier: 7 kernels)"

        def run():
            L.check(lib.swf_basic_block_fwd(C.byref(desc), C.byref(px), C.byref(py), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy),
                                            batch, h, w, ws, wsn, stream))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    n_tok = batch * h * w
    hd = blk.num_heads * blk.dims_per_head
    tbl = (2 * blk.window_size[0] - 1) * (2 * blk.window_size[1] - 1)
    weight_bytes = 2 * (4 * c * hd + 3 * hd + c + tbl + 2 * c) * 4
    alg_bytes = 4 * n_tok * c * BYTES_PER_ELEM + weight_bytes
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC passes (profiles/*_traffic.json: FETCH_SIZE / WRITE_SIZE collected in
    # separate rocprofv3 runs and corrected as MI355X_MICROARCH.md prescribes); only valid for the default workload
    # (PMC counters cannot be read in-process: this field is a constant of the named file, not a measurement of this run)
    traffic, traffic_source = None, None
    tj = os.path.join(REPO, "profiles", TRAFFIC_FILE)
    if os.path.exists(tj) and batch == 16 and size == 256 and not height and not width and c == 24 and precision == "fast" and blk.window_size[0] == 8:
        with open(tj) as f:
            rec = json.load(f)
        traffic = rec.get("hbm_bytes_per_launch")
        traffic_source = (f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of {rec.get('kernel', 'the level-0 block kernel')}, "
                          f"build {rec.get('commit', '?')})")
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kname,
            "ms_per_launch": round(ms, 4), "algorithmic_bytes": alg_bytes, "bytes_per_elem": BYTES_PER_ELEM}


MFMA_PEAK_FLOPS = 2.5e15    # MI355X_MICROARCH.md: dense bf16 / f16 MFMA peak
# kernel symbols of one BasicBlock per level at the default dims with 8x8 / 7x7 windows on 256x256 inputs;
# other shapes dispatch differently (profiles/*_forward_timeline*.txt name what ran)
PMC_LEVELS_THROUGHPUT_FILE = "r03c_pmc_levels_throughput.json"   # levels 2 and 3 in the throughput schedule (the other levels' kernels are the same)
LEVEL_KERNELS_THROUGHPUT = {
    2: "window96_kernel<HID, WS> (four waves per window, two windows per CU)",
    3: "qkv_attn_kernel<192, WS> + mlp_fused_kernel<192, 4, 64> (64-token tiles, the whole hidden range per workgroup)",
}
LEVEL_KERNELS_WIN8 = {
    0: "window24_kernel<HID, WS>", 1: "window48_kernel<HID, WS>", 2: "window96x8_kernel<HID, WS> (maps of > 16 windows: window96_kernel)",
    3: "qkv_attn_kernel<192, WS> + mlp_fused_kernel<192, 4> (+ mlp_reduce_ln_kernel when the hidden dim is split)",
    4: "deep_patch_kernel<384, 192, .., 1152> (Q/K/V) + attn_proj_kernel<WS> + mlp_fused_kernel<384, 8> + mlp_reduce_ln_kernel",
}


def level_rooflines(model, ir, vis, step_ms, iters=5):
    """Per level: HIP-event time of the four BasicBlocks of the stage INSIDE one eager forward (swf_model_forward_profiled: events on
    the launch stream between the stages), algorithmic bytes and flops of one block, and the fractions of the HBM and dense-MFMA
    peaks they amount to.  Algorithmic bytes per block (both streams, e = 4): 4*N*C*e + fp32 weights.  Algorithmic flops
    per block: per stream QKV 6NC^2 + proj 2NC^2 + QK^T 2NtC + P.V 2NtC + MLP 4NC*hid.  The linears run as three
    bf16 MFMAs per product (split-bf16), so the MFMA pipe executes ~3x the linear flops counted here: `frac_mfma` is algorithmic."""
    from swin_unet_image_fusion_amd import _lib as L
    from swin_unet_image_fusion_amd.modules import _ptr, _stream, _workspace
    lib, desc = L.lib(), model._model_desc()
    if desc.precision != L.PREC_FAST:
        return None
    b, _, h, w = ir.shape
    dev = ir.device
    arena = model._get_arena(dev)
    packed = model._get_packed(arena)
    out = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
    ws, wsn = _workspace(lib.swf_model_workspace_bytes(C.byref(desc), b, h, w), dev)
    n = len(model.in_dims_list)
    nseg = 4 * n + 1
    seg = (C.c_float * nseg)()
    acc = [0.0] * nseg
    for it in range(iters + 1):
        L.check(lib.swf_model_forward_profiled(C.byref(desc), _ptr(arena), packed.data_ptr(), _ptr(ir), _ptr(vis), _ptr(out), b, h, w,
                                               ws, wsn, seg, nseg, _stream(dev)))
        if it:   # the first pass warms caches and clocks
            acc = [a + float(v) for a, v in zip(acc, seg)]
    ms = [a / iters for a in acc]
    wh, ww = model.window_size
    mh, mw = model.merging_size
    t = wh * ww
    tbl = (2 * wh - 1) * (2 * ww - 1)
    levels, patches = [], []
    total_bytes = 0
    # MFMA / VALU busy fractions of each level's kernels from the committed counter passes (PMC cannot be read in-process: constants
    # of the named file, collected on the build it names, for the default workload's encoder widths only)
    pmc = None
    pj = os.path.join(REPO, "profiles", PMC_LEVELS_FILE)
    throughput = getattr(model, "schedule", "latency") == "throughput"
    if os.path.exists(pj) and b == 16 and h == 256 and w == 256 and wh == 8 and n == 5:
        with open(pj) as f:
            pmc = json.load(f)
        pmc["file"] = {str(k): PMC_LEVELS_FILE for k in range(n)}
        pt = os.path.join(REPO, "profiles", PMC_LEVELS_THROUGHPUT_FILE)
        if throughput and os.path.exists(pt):
            with open(pt) as f:
                thr = json.load(f)
            for k, v in thr.get("levels", {}).items():
                if v:
                    pmc["levels"][k] = v
                    pmc["file"][k] = PMC_LEVELS_THROUGHPUT_FILE
    hh, wd = h, w
    shapes = []
    for s in range(n):   # map of level s: merged (reflect-padded to the merge size), then padded to a multiple of the window
        hm, wm = -(-hh // mh), -(-wd // mw)
        ho, wo = -(-hm // wh) * wh, -(-wm // ww) * ww
        shapes.append((hh, wd, hm, wm, ho, wo))
        hh, wd = ho, wo
    for side in ("encoder", "decoder"):
        for k in range(n):
            lvl = k if side == "encoder" else n - 1 - k
            c = model.out_dims_list[lvl]
            hid = (model.out_dims_list[lvl] if side == "encoder" else model.in_dims_list[lvl]) * model.mlp_hidden_dims_ratio
            hd = model.att_num_heads * int(model.out_dims_list[lvl] * model.att_dims_per_head_ratio)
            hin, win, hm, wm, ho, wo = shapes[lvl]
            ntok = b * ho * wo
            wbytes = 2 * (3 * c * hd + 3 * hd + hd * c + c + tbl + 4 * c + 2 * c * hid + hid + c) * 4
            blk_bytes = 4 * ntok * c * BYTES_PER_ELEM + wbytes
            blk_flops = 2 * (ntok * (6 * c * hd + 2 * hd * c) + 4 * ntok * t * hd + 4 * ntok * c * hid)
            lin_flops = 2 * (ntok * (6 * c * hd + 2 * hd * c) + 4 * ntok * c * hid)
            seg_i = 2 * lvl + 1 if side == "encoder" else 2 * n + 2 * k
            us = ms[seg_i] * 1e3 / 4
            entry = {"level": lvl, "side": side, "C": c, "hidden": hid, "tokens_per_stream": ntok, "us_per_block": round(us, 2),
                     "kernels": ((LEVEL_KERNELS_THROUGHPUT.get(lvl) if throughput else None) or LEVEL_KERNELS_WIN8.get(lvl, "see profiles/"))
                     if wh in (7, 8) and n == 5 else "see profiles/",
                     "algorithmic_bytes": blk_bytes, "algorithmic_flops": blk_flops,
                     "frac_hbm": round(blk_bytes / (us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4),
                     "frac_mfma": round(blk_flops / (us * 1e-6) / MFMA_PEAK_FLOPS, 4),
                     "frac_mfma_issued": round((blk_flops + 2 * lin_flops) / (us * 1e-6) / MFMA_PEAK_FLOPS, 4)}
            if pmc and side == "encoder" and str(lvl) in pmc.get("levels", {}):
                entry["pmc"] = {"source": f"profiles/{pmc['file'][str(lvl)]} (rocprofv3 --pmc, one block of the level in a loop)",
                                "kernels": {k: {f: v[f] for f in ("us_under_pmc", "mfma_busy", "valu_active", "lds_conflict") if f in v}
                                            for k, v in pmc["levels"][str(lvl)].items() if "split_planes" not in k and "layernorm_vec" not in k}}
            if lvl == 0 and side == "encoder" and c == 24 and hid == 96 and wh == 8:
                # DESIGN.md section 5 issue model of window24_kernel<96, 8>: per wave and window 1 308 VALU wave-instructions priced by issue
                # class (tools/valu_rate_bench.hip) = 2.67 us and 124 MFMAs = 1.76 us, serialised on the wave's SIMD; 4 waves per window
                wave_windows_per_simd = (ntok // t) * 4 / 1024.0
                model_us = (2.67 + 1.76) * wave_windows_per_simd
                entry["issue_model_us"] = round(model_us, 1)
                entry["frac_valu_mfma_issue"] = round(model_us / us, 4)
                entry["bound"] = "VALU + MFMA issue of the SIMD (exp2 and split-bf16 conversions), not HBM"
            levels.append(entry)
            total_bytes += 4 * blk_bytes
            # the patch layer next to the stage: reads 4*Cin (merge) / Cin (un-merge) and writes Cout / 4*Cout per merged token, + skip
            cin = model.in_dims_list[lvl]
            if side == "encoder":
                pbytes = 2 * (b * hm * wm) * (4 * cin + c) * BYTES_PER_ELEM
                pus = ms[2 * lvl] * 1e3
            else:
                pbytes = 2 * (b * hm * wm) * (c + 4 * cin) * BYTES_PER_ELEM + (2 * (b * hin * win) * cin * BYTES_PER_ELEM if lvl > 0 else 0)
                pus = ms[2 * n + 2 * k + 1] * 1e3
            patches.append({"level": lvl, "side": side, "us": round(pus, 2), "algorithmic_bytes": pbytes,
                            "frac_hbm": round(pbytes / (pus * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)})
            total_bytes += pbytes
    head_bytes = 3 * b * h * w * BYTES_PER_ELEM
    head_us = ms[4 * n] * 1e3
    total_bytes += head_bytes
    eager_ms = sum(ms)
    return {"levels": levels, "patch_layers": patches,
            "head": {"us": round(head_us, 2), "algorithmic_bytes": head_bytes, "frac_hbm": round(head_bytes / (head_us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)},
            "forward": {"algorithmic_bytes": total_bytes, "frac_hbm": round(total_bytes / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                        "ms_per_step": round(step_ms, 4), "eager_profiled_ms": round(eager_ms, 4),
                        "note": "whole forward: every fused block reads and writes each stream once (e = 4 B) + weights + patch layers + head, "
                                "divided by the timed step; eager_profiled_ms = sum of the segment events of the per-level pass"},
            "measured_by": f"swf_model_forward_profiled: HIP events on the launch stream between the stages of one eager forward, mean of {iters}"}


def cpu_baseline(cfg, size, pairs, iters):
    """The CPU oracle (port of the reference's PyTorch-CPU forward, pinned to the reference by
    tests/test_oracle_golden.py) on the host cores, bounded sample of the same workload."""
    from oracle import swin_fusion_oracle as O   # bench's cpu_baseline leg is allowed to use the checker
    from swin_unet_image_fusion_amd import MyModel, load_recipe_into, synthetic_pair
    m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(m, seed=0, flavor="default")
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    ir, vis = (torch.from_numpy(a) for a in synthetic_pair(pairs, size, size))
    # the GPU box gives one GPU a 16-core CPU share although os.cpu_count() reports the whole host:
    # more threads than that oversubscribe and slow the baseline down (measured: 128 threads -> 0.41 pairs/s)
    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("SWF_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    with torch.no_grad():
        O.model_forward(sd, cfg, ir[:1], vis[:1])   # warm-up
        best = float("inf")
        for _ in range(iters):
            t = time.perf_counter()
            O.model_forward(sd, cfg, ir, vis)
            best = min(best, time.perf_counter() - t)
    return {"value": round(pairs / best, 3), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{iters} forwards of {pairs} pairs {size}x{size} (min), fp32, torch CPU oracle, {threads} threads; "
                      "the oracle is a port: it ran at ~0.85x the reference's own speed where both could be timed "
                      "(build container, 8 threads: 1.57 vs 1.83 pairs/s)"}


def baseline_config_label(args, cfg, world):
    """Which BASELINE.json config (0-based index into `configs`) this run is, derived from the arguments."""
    w = cfg.window_size[0]
    if args.height or args.width:
        return "not a BASELINE config"
    if cfg.n_levels == 5 and w == 8 and args.size == 256 and args.batch == 16:
        return "BASELINE configs[3], 8 GPUs" if world == 8 else ("BASELINE configs[1]" if world == 1 else f"BASELINE configs[3] shard size on {world} GPUs")
    if cfg.n_levels == 5 and w == 8 and args.size == 512 and args.batch == 16 and world == 1:
        return "BASELINE configs[2]"
    if cfg.n_levels == 5 and w == 16 and args.size == 1024 and args.batch == 8 and world == 1:
        return "BASELINE configs[4]"
    return "not a BASELINE config"


def _rccl_options():
    """RCCL's stream at high priority: its own hardware queue (HIP keeps a queue pool per priority), so the all-gather never sits in a
    queue behind a lane's graph, and its kernel is dispatched ahead of compute when it becomes ready (SWF_RCCL_PRIORITY=0: default stream)."""
    if os.environ.get("SWF_RCCL_PRIORITY", "1") == "0":
        return None
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:   # a torch build without the option: the default stream
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box (tests/test_gpu_bench_two_ranks.py): every rank on device 0, gloo as the transport (RCCL cannot put
    # two ranks on one device).  Everything else — lanes, deferred collectives, barriers, the MAX all-reduce — is the N > 1 path.
    share_gpu = os.environ.get("SWF_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dry = args.dry_run
    if dry:     # the same control flow on CPU tensors over gloo; nothing below touches a GPU or the library
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        sync = torch.cuda.synchronize
    if world == 1 and args.force_collective and not dry:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=_rccl_options())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry or share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev, pg_options=_rccl_options())   # RCCL

    from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
    from swin_unet_image_fusion_amd.shard import ShardedFusion
    cfg = CONFIGS[args.config]
    torch.set_grad_enabled(False)
    if dry:
        model = None
        runner = ShardedFusion(None, world_size=world, rank=rank, use_graph=False, forward_fn=lambda a, b: 0.5 * (a + b))
    else:
        import __graft_entry__ as entry
        if rank == 0:
            entry.build()
        if world > 1:
            dist.barrier()
        model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
        load_recipe_into(model, seed=0, flavor="default")     # random-init weights of the named architecture
        model.to(dev)
        model.precision = args.precision
        runner = ShardedFusion(model, world_size=world, rank=rank, use_graph=not args.no_graph,
                               in_flight=1 if args.no_graph else args.in_flight, force_collective=args.force_collective and world == 1)

    # synthetic IR / visible pairs, distinct per rank, resident in HBM before the timed region
    hh, ww_ = args.height or args.size, args.width or args.size
    ir, vis = synthetic_pair(args.batch, hh, ww_, seed_ir=1 + 2 * rank, seed_vis=2 + 2 * rank)
    ir, vis = torch.from_numpy(ir).to(dev), torch.from_numpy(vis).to(dev)

    for _ in range(max(args.warmup, 1, 0 if dry else runner.in_flight)):   # (every lane of the runner captures and runs once)
        fused = runner.step(ir, vis)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    # every step = forward of the local shard + all-gather of the fused outputs.  Steps are independent batches and the runner keeps
    # `in_flight` of them going: step i+1 is enqueued (its own hipGraph, buffers and stream) before step i is waited for, so its wide
    # level-0 launches run under the latency-bound deep levels of step i, and the gather of step i runs on RCCL's stream under both.
    # Every step is waited for (stream-level) in order; all K forwards and gathers have completed at the final sync.
    pending = []
    for _ in range(args.steps):
        pending.append(runner.step_async(ir, vis))
        if len(pending) >= max(runner.in_flight, 2):
            fused = pending.pop(0).wait()
    while pending:
        fused = pending.pop(0).wait()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert fused.shape[0] == args.batch * world and bool(torch.isfinite(fused).all())
    # the same steps one at a time (each waited for before the next is enqueued): the latency of one step, for the record
    # (its own one-lane runner in the latency schedule: what a caller gets who waits for every step before enqueueing the next)
    serial_ms = None
    if not dry and world == 1 and runner.in_flight > 1 and runner.graph_active:   # (one GPU only: how many lanes a rank found is its own business, and extra collectives on some ranks would hang the job)
        lanes_schedule = model.schedule
        model.schedule = "latency"
        single = ShardedFusion(model, world_size=world, rank=rank, use_graph=True, in_flight=1, force_collective=runner.force_collective)
        for _ in range(2):
            single.step(ir, vis)
        n_serial = min(args.steps, 20)
        sync()
        t1 = time.perf_counter()
        for _ in range(n_serial):
            single.step(ir, vis)
        sync()
        serial_ms = (time.perf_counter() - t1) / n_serial * 1e3
        del single
        model.schedule = lanes_schedule
    # the timed path is the hipGraph replay: check it against one eager forward of the same inputs before reporting it
    graph_equals_eager = None
    if runner.graph_active:
        graph_equals_eager = bool(torch.equal(runner.local_forward(ir, vis), model(ir, vis)))
        assert graph_equals_eager, "hipGraph replay differs from the eager forward"

    if rank == 0:
        total_pairs = args.batch * world * args.steps
        line = {
            "metric": f"fused image-pairs/sec at {hh}x{ww_}, win={cfg.window_size[0]}", "value": round(total_pairs / elapsed, 2),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "fast" else "f32", "data": "synthetic",
            "config": {"workload": f"B={args.batch}/GPU {hh}x{ww_} IR/visible pairs, win={cfg.window_size[0]}, "
                                   f"{cfg.n_levels}-level Swin-UNet fusion forward ({baseline_config_label(args, cfg, world)})",
                       "global_batch": args.batch * world, "precision_mode": args.precision,
                       "arithmetic": "linear layers split-bf16 (bf16x3) MFMA, QK^T and P.V fp16 MFMA (fp32 accumulate), everything else fp32"
                                     if args.precision == "fast" else "exact fp32 (f32-input MFMA)",
                       "residual_stream": "fp32", "hip_graph": runner.graph_active, "graph_equals_eager": graph_equals_eager,
                       "steps_in_flight": runner.in_flight if runner.graph_active else 1,
                       "kernel_schedule": getattr(model, "schedule", None),
                       "one_step_at_a_time": None if serial_ms is None else
                       {"ms_per_step": round(serial_ms, 4), "pairs_per_s": round(args.batch * world / serial_ms * 1e3, 1)},
                       "collective": ("gloo all_gather of GPU tensors (ranks share one GPU: rehearsal)" if share_gpu else
                                      "rccl all_gather_into_tensor of the fused output per step, overlapped with the next steps' forwards") if world > 1 else
                                     ("one-rank rccl all_gather_into_tensor per step (rehearsal)" if args.force_collective else "none"),
                       "weights": "random-init (numpy PCG64 recipe, seed 0)"},
        }
        if dry:
            line["dry_run"] = True
            line["data"] = "synthetic; DRY RUN on CPU over gloo with a stub forward: the value measures nothing"
            line["config"]["collective"] = "gloo all_gather (stub forward)" if world > 1 else "none"
        else:
            line["roofline"] = level0_block_roofline(model, args.batch, args.size, args.precision, height=args.height, width=args.width)
            if not args.no_levels and args.precision == "fast":
                per_level = level_rooflines(model, ir, vis, elapsed / args.steps * 1e3)
                if per_level:
                    line["roofline"].update(per_level)
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(cfg, args.size, args.cpu_pairs, args.cpu_iters)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
