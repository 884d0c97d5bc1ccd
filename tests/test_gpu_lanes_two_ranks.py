"""Two ranks, three lanes each, on the box's one GPU over gloo (tests/lanes_two_ranks_worker.py): the multi-rank ordering of the
overlapped steps' collectives, which a one-rank RCCL group cannot show.  The children are started with subprocess before they touch
the GPU (never an exec of this process); three processes use the card at once."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("lanes", [3, 1])
def test_two_ranks_with_lanes_gather_every_step_in_order(lanes):
    port = str(29800 + os.getpid() % 150 + lanes)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SWF_PARITY_LOG="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "lanes_two_ranks_worker.py"), str(lanes), "8"], cwd=REPO, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} ok" in out, f"rank {rank} rc={p.returncode}\n{out[-2000:]}"
