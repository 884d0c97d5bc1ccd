"""Fast-tier forward against the CPU oracle on odd shapes (ragged maps, small batches, all three window sizes), and repeated forwards
that must equal the first bit for bit.  Sizes keep every reflect pad smaller than its map, as the reference requires (a006:128).

Gates: rel-L2 <= 1e-3 and max|err| <= 1e-3 of max|ref| (north star).  One case (B=1 160x192, seed-3 weights) holds a patch of
ill-conditioned pixels (y 69..71, x 129..132) where the reference's own fp32 answer is 5.3e-6 away from an fp64 evaluation, 67x the
median: a shape that misses the plain max gate is re-checked against the fp64-derived bound of golden_util.close_conditioned
(1e-3*max|ref| + 2^9 x the measured fp32 uncertainty, and only at pixels the fp64 measure calls ill-conditioned)."""
import pytest
import torch
from torch import nn

from oracle import swin_fusion_oracle as O
from swin_unet_image_fusion_amd import CONFIGS, MyModel, load_recipe_into, synthetic_pair
from tests import golden_util as G

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
    assert l2 <= 1e-3, (l2, mx)
    if mx > 1e-3:   # only with fp64 evidence that the offending pixels are ill-conditioned in the reference itself
        model.precision = "fp32"
        exact = model(ir.to(DEV), vis.to(DEV)).cpu()
        ref32, u, pooled, med = G.fp64_uncertainty(O.model_forward, sd, cfg, ir, vis, extra_fp32=(exact,))
        assert torch.equal(ref32, ref)
        _, _, n_ill = G.close_conditioned(got, ref, pooled, med, 1e-3, 1e-3)
        print(f"{cfg_name} b{b} {h}x{w}: max-rel {mx:.2e} on {n_ill} ill-conditioned pixels (fp32-vs-fp64 {float(u.max()):.2e}, median {med:.2e})")


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
