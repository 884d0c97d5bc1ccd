"""Training side, first stage: the backward of one BasicBlock (swf_basic_block_bwd, exact fp32) against
torch.autograd of the CPU oracle's basic_block on the same weights and inputs — input gradients and every parameter gradient
(LayerNorm affine, Q/K/V/projection weights and biases, the relative-position bias table, both MLP layers), both streams.  The
reference has no backward code of its own: autograd of its forward IS its backward (a016_train.py:150-196), and the oracle is that
forward restated (pinned to the reference by tests/test_oracle_golden.py)."""
import pytest
import torch
from torch import nn

import __graft_entry__ as entry
from oracle import swin_fusion_oracle as O
from swin_unet_image_fusion_amd import (CONFIGS, BasicBlock, MyModel, MyPadding, PatchMergingAndLinearLayer, StateRecorder, load_recipe_into,
                                        synthetic_pair)
from tests import golden_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

_CASES = [  # C, heads, d, win, hidden, (B,H,W), shift, cross, dual
    (8, 2, 4, 4, 32, (2, 8, 12), True, True, True),
    (8, 2, 4, 4, 12, (1, 8, 8), False, False, True),
    (24, 8, 3, 8, 96, (1, 16, 16), True, False, True),
    (24, 8, 3, 8, 4, (1, 8, 16), True, True, True),          # decoder width, edge windows only
    (12, 4, 3, 7, 24, (1, 14, 14), True, True, True),         # 7x7 windows
    (16, 2, 8, 4, 40, (2, 8, 8), True, False, False),         # single-path block
    (48, 8, 6, 8, 192, (1, 8, 8), True, True, True),          # one window per map: the shift mask covers most of the score tile
    (8, 2, 4, 16, 16, (1, 16, 32), True, True, True),         # 16x16 windows: the probability tile does not fit in LDS (recompute kernel)
]


@pytest.fixture(scope="module", autouse=True)
def _built():
    entry.build()
    yield


@pytest.mark.parametrize("case", _CASES, ids=[f"C{c[0]}_w{c[3]}_hid{c[4]}_s{int(c[6])}c{int(c[7])}d{int(c[8])}" for c in _CASES])
def test_basic_block_backward_vs_autograd_of_the_oracle(case):
    C_, nh, d, win, hid, (b, h, w), shift, cross, dual = case
    ref_dual = BasicBlock(C_, nh, d, (win, win), shift, True, cross and dual, True, 0.0, 0.0, hid, nn.ELU(inplace=True), 0.0).eval()
    load_recipe_into(ref_dual, seed=41, flavor="stress")
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref_dual.state_dict().items()}
    x = G.randn((b, C_, h, w), 801).requires_grad_(True)
    y = G.randn((b, C_, h, w), 802).requires_grad_(True)
    wx, wy = G.randn((b, C_, h, w), 803), G.randn((b, C_, h, w), 804)     # a generic linear functional of the outputs as the loss
    ox, oy = O.basic_block(sd, "", x, y, cross=cross and dual, shift=shift, num_heads=nh, dims_per_head=d, window_size=(win, win))
    loss = (ox * wx).sum() + ((oy * wy).sum() if dual else 0.0)
    loss.backward()

    m = BasicBlock(C_, nh, d, (win, win), shift, dual, cross and dual, True, 0.0, 0.0, hid, nn.ELU(inplace=True), 0.0).eval()
    m.load_state_dict({k: v.detach() for k, v in sd.items() if k in m.state_dict()})
    m.to(DEV)
    m.precision = "fp32"
    xg = x.detach().to(DEV).requires_grad_(True)
    yg = y.detach().to(DEV).requires_grad_(True) if dual else None
    out = m(xg, yg) if dual else m(xg)
    if dual:
        (out[0] * wx.to(DEV)).sum().add((out[1] * wy.to(DEV)).sum()).backward()
    else:
        o0 = out[0] if isinstance(out, tuple) else out
        (o0 * wx.to(DEV)).sum().backward()

    def close(got, ref, what, scale):
        got, ref = got.detach().cpu().double(), ref.detach().double()
        err = float((got - ref).abs().max())
        assert err <= 2e-4 * max(float(ref.abs().max()), scale), (what, err, float(ref.abs().max()), scale)

    close(xg.grad, x.grad, "dL/dx", 0.0)
    if dual:
        close(yg.grad, y.grad, "dL/dy", 0.0)
    named = dict(m.named_parameters())
    assert named, "no parameters"
    gscale = max(float(sd[k].grad.abs().max()) for k in named if sd[k].grad is not None)
    checked = 0
    for k, p in named.items():
        assert p.grad is not None, k
        # (gradients that vanish identically — a key bias shifts every score of a query alike — are compared on the scale of the largest)
        close(p.grad, sd[k].grad, k, 1e-2 * gscale)
        checked += 1
    assert checked >= (17 if not dual else 34) - 2


def test_backward_is_bit_reproducible():
    """Every sum over tokens runs in a fixed order (chunked partial sums + ordered reduce, no atomics)."""
    m = BasicBlock(24, 8, 3, (8, 8), True, True, True, True, 0.0, 0.0, 96, nn.ELU(inplace=True), 0.0).eval()
    load_recipe_into(m, seed=42, flavor="stress")
    m.to(DEV)
    m.precision = "fp32"
    x0, y0 = G.randn((2, 24, 16, 24), 811).to(DEV), G.randn((2, 24, 16, 24), 812).to(DEV)
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
        ox, oy = m(x, y)
        (ox.square().sum() + oy.sum()).backward()
        grads.append([x.grad.clone(), y.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))


@pytest.mark.parametrize("enc,cin,cout,shape", [(True, 1, 24, (2, 16, 12)), (True, 8, 16, (1, 8, 8)), (False, 16, 8, (1, 4, 6)), (False, 24, 1, (2, 8, 8)),
                                                (True, 96, 192, (1, 4, 4)), (False, 384, 192, (1, 2, 2))])
def test_patch_layer_backward_vs_autograd_of_the_oracle(enc, cin, cout, shape):
    """PatchMergingAndLinearLayer (a011:244-264) under autograd: input and parameter gradients of both streams."""
    b, h, w = shape
    m = PatchMergingAndLinearLayer(belongs_to_encoder=enc, use_dual_path=True, in_dims=cin, out_dims=cout, patch_merging_size_recorder=StateRecorder(),
                                   merging_or_unmerging_size=(2, 2), activation_func=nn.ELU(inplace=True)).eval()
    load_recipe_into(m, seed=43, flavor="stress")
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    x, y = G.randn((b, cin, h, w), 821).requires_grad_(True), G.randn((b, cin, h, w), 822).requires_grad_(True)
    ox, oy = O.patch_layer(sd, "", x, y, encoder=enc, merging_size=(2, 2))
    wx, wy = G.randn(tuple(ox.shape), 823), G.randn(tuple(oy.shape), 824)
    ((ox * wx).sum() + (oy * wy).sum()).backward()
    m.to(DEV)
    xg, yg = x.detach().to(DEV).requires_grad_(True), y.detach().to(DEV).requires_grad_(True)
    gx, gy = m(xg, yg)
    ((gx * wx.to(DEV)).sum() + (gy * wy.to(DEV)).sum()).backward()

    def close(got, ref, what):
        got, ref = got.detach().cpu().double(), ref.detach().double()
        assert float((got - ref).abs().max()) <= 2e-4 * max(float(ref.abs().max()), 1e-3), (what, float((got - ref).abs().max()), float(ref.abs().max()))

    close(xg.grad, x.grad, "dL/dx"); close(yg.grad, y.grad, "dL/dy")
    for k, p in m.named_parameters():
        close(p.grad, sd[k].grad, k)


@pytest.mark.parametrize("win,shape", [((7, 7), (1, 3, 10, 13)), ((2, 2), (2, 2, 5, 4)), ((8, 8), (1, 2, 16, 16)), ((4, 4), (1, 1, 6, 7))])
def test_padding_backward_vs_autograd_of_the_oracle(win, shape):
    """MyPadding (a006:122-146) under autograd: reflect pad then crop, and the pad alone."""
    fr, pr = StateRecorder(), StateRecorder()
    enc, dec = MyPadding(True, win, True, fr, pr).eval(), MyPadding(False, win, True, fr, pr).eval()
    x, y = G.randn(shape, 831).requires_grad_(True), G.randn(shape, 832).requires_grad_(True)
    px, pad = O.pad_to_multiple(x, win)
    py, _ = O.pad_to_multiple(y, win)
    wx, wy = G.randn(tuple(px.shape), 833), G.randn(tuple(py.shape), 834)
    ((px * wx).sum() + (py * wy).sum() + O.crop_padding(px * 2.0, pad).sum()).backward()
    xg, yg = x.detach().to(DEV).requires_grad_(True), y.detach().to(DEV).requires_grad_(True)
    gx, gy = enc(xg, yg)
    cx, _ = dec(gx * 2.0, gy)
    ((gx * wx.to(DEV)).sum() + (gy * wy.to(DEV)).sum() + cx.sum()).backward()
    assert torch.allclose(xg.grad.cpu(), x.grad, rtol=1e-5, atol=1e-6) and torch.allclose(yg.grad.cpu(), y.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cfg_name,shape", [("tiny", (2, 16, 16)), ("tiny", (1, 18, 22)), ("tiny7", (1, 40, 36)), ("win8_4stage", (1, 128, 128))])
def test_whole_model_backward_vs_autograd_of_the_oracle(cfg_name, shape):
    """MyModel.forward under torch.autograd (module by module, a013:209-230; eval() semantics): dL/d(ir), dL/d(vis) and the gradient of
    EVERY parameter of the model (all stages' blocks, patch layers with padding, skip adds, final head with BatchNorm affine) against
    autograd of the oracle's model_forward; the loss is a generic linear functional plus an L1 term like the reference's (a008:226-282
    mixes L1 / SSIM / gradient terms; kornia, which it needs, is absent here)."""
    cfg = CONFIGS[cfg_name]
    b, h, w = shape
    m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    load_recipe_into(m, seed=7, flavor="stress")
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in m.state_dict().items()}
    ir, vis = (torch.from_numpy(a) for a in synthetic_pair(b, h, w, seed_ir=51, seed_vis=52))
    ir.requires_grad_(True); vis.requires_grad_(True)
    wgt, tgt = G.randn((b, 1, h, w), 851), G.randn((b, 1, h, w), 852) * 0.1
    out = O.model_forward(sd, cfg, ir, vis)
    ((out * wgt).sum() + (out - tgt).abs().sum()).backward()
    m.to(DEV)
    irg, visg = ir.detach().to(DEV).requires_grad_(True), vis.detach().to(DEV).requires_grad_(True)
    outg = m(irg, visg)
    assert outg.requires_grad
    fwd_err = float((outg.detach().cpu() - out.detach()).abs().max() / out.detach().abs().max())
    assert fwd_err <= 2e-3, fwd_err          # (the blocks' forward runs in the model's tier; the backward recomputes in exact fp32)
    ((outg * wgt.to(DEV)).sum() + (outg - tgt.to(DEV)).abs().sum()).backward()

    def rel(got, ref):
        got, ref = got.detach().cpu().double(), ref.detach().double()
        return float((got - ref).norm() / ref.norm().clamp_min(1e-30))

    assert rel(irg.grad, ir.grad) <= 2e-3 and rel(visg.grad, vis.grad) <= 2e-3, (rel(irg.grad, ir.grad), rel(visg.grad, vis.grad))
    worst, n_checked = 0.0, 0
    gmax = max(float(v.grad.abs().max()) for v in sd.values() if v.requires_grad and v.grad is not None)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        got, ref = p.grad.detach().cpu().double(), sd[k].grad.detach().double()
        err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * gmax)
        worst = max(worst, err)
        n_checked += 1
    assert worst <= 5e-3, worst
    assert n_checked == len(list(m.parameters()))


@pytest.mark.parametrize("cfg_name,shape", [("tiny", (3, 16, 16)), ("tiny7", (2, 30, 26)), ("win8_4stage", (2, 128, 128))])
def test_training_mode_steps_vs_autograd_of_the_oracle(cfg_name, shape):
    """model.train() as the reference trains (a016:137): the head's BatchNorm2d normalises with the batch statistics (whose gradient
    flows back into conv1 and the decoder), and its running statistics move by momentum 0.1 with the unbiased variance.  Three plain
    SGD steps on the GPU model against three steps of autograd on the oracle from the same start: output, every gradient of the first
    step, the running statistics and the weights after the last step."""
    cfg = CONFIGS[cfg_name]
    b, h, w = shape
    m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True)))
    load_recipe_into(m, seed=11, flavor="stress")
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in m.state_dict().items()}
    m.to(DEV).eval()
    with torch.no_grad():      # a fused forward before the training steps: its weight arena holds the STARTING weights
        m(*(torch.from_numpy(a).to(DEV) for a in synthetic_pair(b, h, w, seed_ir=60, seed_vis=70)))
    assert m._arena is not None
    m.train()
    lr, first = 1e-3, None
    for step in range(3):
        ir, vis = (torch.from_numpy(a) for a in synthetic_pair(b, h, w, seed_ir=61 + step, seed_vis=71 + step))
        tgt = torch.maximum(ir, vis)
        for v in sd.values():
            v.grad = None
        out = O.model_forward(sd, cfg, ir, vis, training=True)
        (out - tgt).square().mean().backward()     # (smooth: an L1 term flips sign where the two forwards straddle the target)
        m.zero_grad(set_to_none=True)
        outg = m(ir.to(DEV), vis.to(DEV))
        (outg - tgt.to(DEV)).square().mean().backward()
        assert float((outg.detach().cpu() - out.detach()).abs().max() / out.detach().abs().max()) <= 2e-3
        if step == 0:
            gmax = max(float(v.grad.abs().max()) for v in sd.values() if v.requires_grad and v.grad is not None)
            first = max(float((p.grad.cpu().double() - sd[k].grad.double()).abs().max()) / max(float(sd[k].grad.abs().max()), 1e-3 * gmax)
                        for k, p in m.named_parameters())
            assert first <= 5e-3, first
        with torch.no_grad():
            for k, p in m.named_parameters():
                p -= lr * p.grad
                sd[k] -= lr * sd[k].grad
    # a016:202: the evaluation forward after the training steps (fused, no grad) must run the UPDATED weights without a manual refresh
    m.eval()
    with torch.no_grad():
        ev = m(ir.to(DEV), vis.to(DEV))
    ev_ref = O.model_forward({k: v.detach() for k, v in sd.items()}, cfg, ir, vis)
    assert float((ev.cpu() - ev_ref).abs().max() / ev_ref.abs().max()) <= 2e-3
    fresh = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True)))
    fresh.load_state_dict(m.state_dict(), strict=True)
    fresh.to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(fresh(ir.to(DEV), vis.to(DEV)), ev)      # bit for bit what a model freshly loaded with the trained weights gives
    m.train()
    got = m.state_dict()
    for k in ("final_layer.1.running_mean", "final_layer.1.running_var"):
        assert torch.allclose(got[k].cpu(), sd[k], rtol=2e-3, atol=1e-6), (k, got[k].cpu(), sd[k])
    assert int(got["final_layer.1.num_batches_tracked"]) == 3
    for k, p in m.named_parameters():
        ref = sd[k].detach()
        assert float((p.detach().cpu() - ref).abs().max()) <= 1e-3 * lr * max(gmax, 1.0) + 1e-5 * float(ref.abs().max()), k


def test_head_batch_statistics_match_torch():
    """swf_final_head_batch_stats alone: mean / biased variance of conv1's output and the running-statistics update, against
    F.batch_norm(training=True) on the oracle's conv1."""
    import torch.nn.functional as F
    cfg = CONFIGS["tiny"]
    m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True)))
    load_recipe_into(m, seed=3, flavor="stress")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, y = G.randn((2, 1, 11, 9), 901), G.randn((2, 1, 11, 9), 902)
    z = F.conv2d(F.pad(torch.cat([x, y], 1), (1, 1, 1, 1), mode="reflect"), sd["final_layer.0.weight"], sd["final_layer.0.bias"])
    mean, var = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    rm, rv = sd["final_layer.1.running_mean"].clone(), sd["final_layer.1.running_var"].clone()
    rm0 = rm.clone()
    F.batch_norm(z, rm, rv, None, None, training=True, momentum=0.1)
    m.to(DEV).train()
    xg, yg = x.to(DEV).requires_grad_(True), y.to(DEV).requires_grad_(True)
    out = m.do_final_layer(xg, yg)
    ref = O.final_head(sd, x, y, 3, training=True)
    assert torch.allclose(out.detach().cpu(), ref, rtol=1e-4, atol=1e-5)
    bn = m.final_layer[1]
    assert torch.allclose(bn.running_mean.cpu(), rm, rtol=1e-5, atol=1e-6) and torch.allclose(bn.running_var.cpu(), rv, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_mean.cpu(), 0.9 * rm0 + 0.1 * mean, rtol=1e-5, atol=1e-6)


_WA_CASES = [  # C, heads, d, win, (B,H,W), shift, mode (self: q=k=v one tensor; cross: k=v the other stream; mixed: three tensors), qkv bias
    (24, 8, 3, 8, (1, 16, 16), True, "self", True),
    (24, 8, 3, 8, (2, 8, 16), False, "cross", True),
    (12, 4, 3, 7, (1, 14, 7), True, "cross", True),
    (16, 2, 5, 4, (1, 8, 8), True, "mixed", False),          # heads * d != C, no q/k/v bias
    (8, 2, 4, 16, (1, 16, 16), True, "self", True),          # 16x16 window: the recompute kernel
]


@pytest.mark.parametrize("case", _WA_CASES, ids=[f"C{c[0]}_w{c[3]}_s{int(c[5])}_{c[6]}" for c in _WA_CASES])
def test_window_attention_backward_vs_autograd_of_the_oracle(case):
    """WindowAttention on its own (a001:448-474, the drop-in the north star names) under torch.autograd: input gradients — summed by
    autograd where one tensor is passed as several of q, k, v — and every parameter gradient against autograd of the oracle."""
    from swin_unet_image_fusion_amd import WindowAttention
    c, heads, d, win, (b, h, w), shift, mode, bias = case
    m = WindowAttention(c, heads, d, (win, win), shift, mode != "self", bias, 0.0, 0.0).eval()
    load_recipe_into(m, seed=21, flavor="stress")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    a, bt, ct = (G.randn((b, c, h, w), 700 + i).requires_grad_(True) for i in range(3))
    pick = {"self": (a, a, a), "cross": (a, bt, bt), "mixed": (a, bt, ct)}[mode]
    wgt = G.randn((b, c, h, w), 710)
    out = O.window_attention(sd, "", *pick, num_heads=heads, dims_per_head=d, window_size=(win, win), use_cyclic_shift=shift)
    (out * wgt).sum().backward()
    m.to(DEV)
    ag, bg, cg = (t.detach().to(DEV).requires_grad_(True) for t in (a, bt, ct))
    pickg = {"self": (ag, ag, ag), "cross": (ag, bg, bg), "mixed": (ag, bg, cg)}[mode]
    outg = m(*pickg)
    assert outg.requires_grad
    assert float((outg.detach().cpu() - out.detach()).abs().max() / out.detach().abs().max()) <= 2e-3
    (outg * wgt.to(DEV)).sum().backward()
    rel = lambda got, ref: float((got.detach().cpu().double() - ref.detach().double()).norm() / ref.detach().double().norm().clamp_min(1e-30))
    used = {"self": [(ag, a)], "cross": [(ag, a), (bg, bt)], "mixed": [(ag, a), (bg, bt), (cg, ct)]}[mode]
    for got, ref in used:
        assert rel(got.grad, ref.grad) <= 2e-3, rel(got.grad, ref.grad)
    # (the key bias has NO gradient mathematically — a shift of every key by b_k moves all scores of a query by the same q.b_k —, so both
    #  sides hold rounding noise there: errors are measured against the largest gradient of the module, not against each tensor's own size)
    gmax = max(float(v.grad.abs().max()) for v in sd.values())
    for k, p in m.named_parameters():
        assert p.grad is not None and sd[k].grad is not None, k
        err = float((p.grad.detach().cpu().double() - sd[k].grad.double()).abs().max()) / max(float(sd[k].grad.abs().max()), 1e-3 * gmax)
        assert err <= 5e-3, (k, err)


@pytest.mark.parametrize("dual", [True, False], ids=["dual", "single"])
def test_inner_modules_backward_compose_like_the_block(dual):
    """The reference's BasicBlock assembled from this package's inner modules (a005:70-82: AddAndLayerNorm around AutoPathWinAtt, then
    around AutoPathMLP) under torch.autograd — the caller that swaps only a001-a004: gradients of both inputs and of every parameter
    equal autograd of the oracle's basic_block, and equal the fused block's own backward."""
    from swin_unet_image_fusion_amd import AddAndLayerNormWithOtherModule
    c, heads, d, win, hid, (b, h, w) = 24, 8, 3, 8, 96, (1, 16, 16)
    m = BasicBlock(c, heads, d, (win, win), True, dual, dual, True, 0.0, 0.0, hid, nn.ELU(inplace=True), 0.0).eval()
    load_recipe_into(m, seed=5, flavor="stress")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, y = G.randn((b, c, h, w), 801).requires_grad_(True), G.randn((b, c, h, w), 802).requires_grad_(True)
    wx, wy = G.randn((b, c, h, w), 803), G.randn((b, c, h, w), 804)
    if dual:
        ox, oy = O.basic_block(sd, "", x, y, cross=True, shift=True, num_heads=heads, dims_per_head=d, window_size=(win, win))
        ((ox * wx).sum() + (oy * wy).sum()).backward()
    else:   # single path (a005 with use_dual_path False): the oracle's pieces by hand
        import torch.nn.functional as F
        n1 = O.layer_norm_channels(x, sd["stage_1.norm_layer_1.weight"], sd["stage_1.norm_layer_1.bias"])
        a1 = O.window_attention(sd, "auto_path_win_att.window_attention_x.", n1, n1, n1, num_heads=heads, dims_per_head=d,
                                window_size=(win, win), use_cyclic_shift=True)
        x1r = x + a1
        n2 = O.layer_norm_channels(x1r, sd["stage_2.norm_layer_1.weight"], sd["stage_2.norm_layer_1.bias"])
        hdn = F.elu(F.conv2d(n2, sd["auto_path_mlp.mlp_x_1.weight"], sd["auto_path_mlp.mlp_x_1.bias"]))
        ox = x1r + F.conv2d(hdn, sd["auto_path_mlp.mlp_x_2.weight"], sd["auto_path_mlp.mlp_x_2.bias"])
        (ox * wx).sum().backward()
    m.to(DEV)
    xg, yg = x.detach().to(DEV).requires_grad_(True), y.detach().to(DEV).requires_grad_(True)
    assert isinstance(m.stage_1, AddAndLayerNormWithOtherModule)
    rel = lambda got, ref: float((got.detach().cpu().double() - ref.detach().double()).norm() / ref.detach().double().norm().clamp_min(1e-30))
    if dual:
        x1, y1 = m.stage_1(xg, yg)
        x2, y2 = m.stage_2(x1, y1)
        assert x2.requires_grad and y2.requires_grad
        ((x2 * wx.to(DEV)).sum() + (y2 * wy.to(DEV)).sum()).backward()
        assert rel(xg.grad, x.grad) <= 2e-3 and rel(yg.grad, y.grad) <= 2e-3
    else:
        x2 = m.stage_2(m.stage_1(xg))
        assert x2.requires_grad
        (x2 * wx.to(DEV)).sum().backward()
        assert rel(xg.grad, x.grad) <= 2e-3
    seen = 0
    gmax = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, p in m.named_parameters():
        if sd[k].grad is None:
            continue
        assert p.grad is not None, k
        err = float((p.grad.detach().cpu().double() - sd[k].grad.double()).abs().max()) / max(float(sd[k].grad.abs().max()), 1e-3 * gmax)
        assert err <= 5e-3, (k, err)
        seen += 1
    assert seen >= (26 if dual else 13)
