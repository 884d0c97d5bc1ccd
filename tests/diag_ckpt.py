"""Where does the fast tier differ most from the oracle on the stress-weight 4-stage model? (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import __graft_entry__ as entry
entry.build()
from oracle import swin_fusion_oracle as O
from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
torch.set_grad_enabled(False)
cfg = CONFIGS["win8_4stage"]
m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
load_recipe_into(m, seed=17, flavor="stress")
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
ir, vis = (torch.from_numpy(a) for a in synthetic_pair(2, 128, 128, seed_ir=21, seed_vis=22))
ref = O.model_forward(sd, cfg, ir, vis)
m.to("cuda:0")
for prec in ("fp32", "fast"):
    m.precision = prec
    out = m(ir.cuda(), vis.cuda()).cpu()
    err = (out - ref).abs()
    print(prec, "rel-l2", float((out - ref).norm() / ref.norm()), "max-rel", float(err.max() / ref.abs().max()), "ref max", float(ref.abs().max()))
    top = torch.topk(err.flatten(), 8)
    for v, i in zip(top.values.tolist(), top.indices.tolist()):
        b, rem = divmod(i, 128 * 128); y, x = divmod(rem, 128)
        print(f"   err {v:.3e} at b={b} y={y} x={x}  out={float(out[b,0,y,x]):.5f} ref={float(ref[b,0,y,x]):.5f}")
    print("   err quantiles", [float(torch.quantile(err.flatten(), q)) for q in (0.5, 0.9, 0.99, 0.999, 0.9999)])
