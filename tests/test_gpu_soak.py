"""Fast-tier forward against the CPU oracle on odd shapes (ragged maps, small batches, all three window sizes), and repeated forwards
that must equal the first bit for bit.  Sizes keep every reflect pad smaller than its map, as the reference requires (a006:128).

Last build: rel-L2 3.3e-5 ... 1.4e-4 on the nine shapes.  The largest max-error (1.3e-3 of max|ref| at B=1 160x192, seed-3 weights) sits
on one ill-conditioned pixel where the exact fp32 tier also has its largest error (5e-6, 50x its median); the same case measured
1.5e-3 with the kernels of the start of round 2 — hence the 5e-3 max gate here, next to the north star's 1e-3 on rel-L2."""
import pytest
import torch
from torch import nn

from oracle import swin_fusion_oracle as O
from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

_SHAPES = [("win8", 1, 160, 192), ("win8", 3, 168, 200), ("win8", 2, 264, 248), ("win8", 5, 256, 256), ("win7", 2, 130, 150),
           ("win7", 1, 224, 224), ("win7", 3, 150, 134), ("win16", 1, 288, 304), ("win16", 2, 512, 512)]


@pytest.mark.parametrize("cfg_name,b,h,w", _SHAPES, ids=[f"{c}_b{b}_{h}x{w}" for c, b, h, w in _SHAPES])
def test_odd_shapes_fast_tier_vs_oracle(cfg_name, b, h, w):
    torch.set_grad_enabled(False)
    cfg = CONFIGS[cfg_name]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=3, flavor="default")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ir, vis = (torch.from_numpy(a) for a in synthetic_pair(b, h, w, seed_ir=31, seed_vis=32))
    ref = O.model_forward(sd, cfg, ir, vis)
    model.to(DEV)
    model.precision = "fast"
    got = model(ir.to(DEV), vis.to(DEV)).cpu()
    l2 = float((got - ref).norm() / ref.norm())
    mx = float((got - ref).abs().max() / ref.abs().max())
    assert l2 <= 1e-3 and mx <= 5e-3, (l2, mx)


@pytest.mark.parametrize("cfg_name,b,size", [("win8", 16, 256), ("win7", 16, 224), ("win8", 5, 512), ("win16", 2, 1024), ("win8", 3, 320)])
def test_repeated_forwards_are_bit_identical(cfg_name, b, size):
    """A data race in a kernel (LDS exchange buffers laid over live images, in-place cross blocks) would show as a difference."""
    torch.set_grad_enabled(False)
    cfg = CONFIGS[cfg_name]
    model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(model, seed=1, flavor="stress")
    model.to(DEV)
    model.precision = "fast"
    ir, vis = (torch.from_numpy(a).to(DEV) for a in synthetic_pair(b, size, size))
    ref = model(ir, vis).clone()
    assert bool(torch.isfinite(ref).all())
    for _ in range(30):
        assert torch.equal(model(ir, vis), ref)
