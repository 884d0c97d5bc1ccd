"""Fast-tier forward against the CPU oracle on odd shapes (ragged maps, small batches, all three window sizes): rel-L2 and max error
relative to max|ref| per case.  The oracle (oracle/, test infrastructure) is only the checker.    python tools/soak_parity.py

Last build: rel-L2 3.3e-5 ... 1.4e-4 on nine shapes.  The largest max-error (1.3e-3 of max|ref| at B=1 160x192, seed-3 weights) sits on one
ill-conditioned pixel where the exact fp32 tier also has its largest error (5e-6, 50x its median); the same case measured 1.5e-3
with the kernels of the start of the round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import __graft_entry__ as entry
entry.build()
from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
from oracle import swin_fusion_oracle as O
torch.set_grad_enabled(False)
torch.set_num_threads(16)
# (sizes keep every reflect pad smaller than its map, as the reference requires: a006:128)
cases = [("win8", 1, 160, 192), ("win8", 3, 168, 200), ("win8", 2, 264, 248), ("win8", 5, 256, 256), ("win7", 2, 130, 150),
         ("win7", 1, 224, 224), ("win7", 3, 150, 134), ("win16", 1, 288, 304), ("win16", 2, 512, 512)]
worst = 0.0
for cfg_name, b, h, w in cases:
    cfg = CONFIGS[cfg_name]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=3, flavor="default")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ir, vis = (torch.from_numpy(a) for a in synthetic_pair(b, h, w, seed_ir=31, seed_vis=32))
    ref = O.model_forward(sd, cfg, ir, vis)
    model.to("cuda:0")
    model.precision = "fast"
    got = model(ir.cuda(), vis.cuda()).cpu()
    l2 = float((got - ref).norm() / ref.norm())
    mx = float((got - ref).abs().max() / ref.abs().max())
    worst = max(worst, l2)
    print(f"{cfg_name} B={b} {h}x{w}: rel-L2 {l2:.2e}  max/max|ref| {mx:.2e}", flush=True)
print(f"worst rel-L2 {worst:.2e} (north star: 1e-3)")
sys.exit(0 if worst < 1e-3 else 1)
