"""bench.py's N > 1 control flow (process-group init, barriers, the pipelined step_async loop, the MAX all-reduce of the elapsed
time, one JSON line from rank 0) rehearsed on CPU: two gloo ranks, CPU tensors, a stub forward (`--dry-run`).  No scaling number
can come of it; it makes sure the driver's first multi-GPU lease is not the first execution of that code."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout     # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_two_rank_dry_run_prints_one_line_from_rank_zero():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--batch", "3", "--size", "32", "--dry-run"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2
    assert d["config"]["global_batch"] == 6 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 6 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-2
    assert "roofline" not in d and "cpu_baseline" not in d     # nothing measured


def test_single_process_dry_run():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "16", "--dry-run"],
                       cwd=REPO, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["dry_run"] is True and d["n_gpus"] == 1 and d["config"]["global_batch"] == 2
