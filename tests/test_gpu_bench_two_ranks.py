"""bench.py as the driver launches it for N = 2 (python -m torch.distributed.run ... bench.py --gpus 2), rehearsed on the box's one
GPU: SWF_BENCH_SHARE_GPU=1 puts both ranks on device 0 with gloo as the transport.  The real model, three lanes per rank, deferred
all-gathers, barriers and the MAX all-reduce of the elapsed time all run; only RCCL itself is not on this path.  The child is started
with subprocess (never an exec of this process); three processes use the card."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_the_gpu():
    env = dict(os.environ, SWF_BENCH_SHARE_GPU="1", SWF_PARITY_LOG="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = str(29600 + os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "3", "--batch", "4", "--size", "128", "--config", "win8_4stage",
           "--no-levels", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]          # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 8 and d["config"]["hip_graph"] and d["config"]["graph_equals_eager"]
    assert d["config"]["steps_in_flight"] >= 2 and "gloo" in d["config"]["collective"]
    assert abs(d["value"] - 8 * 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-3
