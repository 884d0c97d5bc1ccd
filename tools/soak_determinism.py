"""Run the fast-tier forward many times on the same inputs and check that every output equals the first bit for bit (a data race in a
kernel would show up as a difference).    python tools/soak_determinism.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import __graft_entry__ as entry
entry.build()
from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
torch.set_grad_enabled(False)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for cfg_name, batch, size in (("win8", 16, 256), ("win7", 16, 224), ("win8", 5, 512), ("win16", 2, 1024), ("win8", 3, 320)):
    cfg = CONFIGS[cfg_name]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=1, flavor="stress")
    model.to("cuda:0")
    model.precision = "fast"
    ir, vis = synthetic_pair(batch, size, size)
    ir, vis = torch.from_numpy(ir).cuda(), torch.from_numpy(vis).cuda()
    ref = model(ir, vis).clone()
    bad = 0
    for _ in range(iters):
        if not torch.equal(model(ir, vis), ref): bad += 1
    torch.cuda.synchronize()
    print(f"{cfg_name} B={batch} {size}x{size}: {iters} forwards, {bad} differ from the first, finite {bool(torch.isfinite(ref).all())}", flush=True)
