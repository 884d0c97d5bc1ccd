"""Worker of tests/test_gpu_lanes_two_ranks.py: one of two ranks that SHARE the box's one GPU, gloo as the process group (RCCL cannot
put two ranks on one device).  Everything but the transport is the multi-GPU path: three hipGraph lanes per rank, staging ring,
collectives started in_flight - 1 steps late and oldest first, handles waited for in order.  Every gathered result must equal the
eager forwards of BOTH ranks' batches of that step, bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from torch import nn


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lanes, steps = int(sys.argv[1]), int(sys.argv[2])
    from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
    from swin_unet_image_fusion_amd.shard import ShardedFusion
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = CONFIGS["win8_4stage"]
    m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(m, seed=0, flavor="default")
    m.to(dev)
    runner = ShardedFusion(m, world_size=world, rank=rank, use_graph=True, in_flight=lanes)
    batch = lambda r, i: tuple(torch.from_numpy(a).to(dev) for a in synthetic_pair(2, 128, 128, 100 + 10 * i + r, 200 + 10 * i + r))
    mine = [batch(rank, i) for i in range(steps)]
    got, pending = [], []
    for ir, vis in mine:
        pending.append(runner.step_async(ir, vis))
        if len(pending) == lanes:
            got.append(pending.pop(0).wait().clone())
    while pending:
        got.append(pending.pop(0).wait().clone())
    torch.cuda.synchronize()
    assert not runner._unissued and runner.captures == runner.in_flight
    for i, g in enumerate(got):
        want = torch.cat([m(*batch(r, i)) for r in range(world)])
        assert g.shape == want.shape and torch.equal(g, want), f"rank {rank} step {i}"
    # a rank that waits out of step with the other: the collectives are still issued oldest first
    h1, h2 = runner.step_async(*mine[0]), runner.step_async(*mine[1])
    if rank == 0:
        a, b = h1.wait().clone(), h2.wait().clone()
    else:
        b = h2.wait().clone()
        a = h1.wait().clone()
    assert torch.equal(a, got[0]) and torch.equal(b, got[1])
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok lanes={runner.in_flight}", flush=True)


if __name__ == "__main__":
    main()
