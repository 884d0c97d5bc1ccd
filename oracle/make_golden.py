"""Generate tests/golden/* by running the REAL reference on CPU (build container only).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference (/root/reference, read-only) is imported, never copied: only *data* leaves —
expected outputs, the key/shape/alias table of its state_dict, and the case descriptions.
Weights and inputs are NOT stored: they come from the deterministic recipes in
`swin_unet_image_fusion_amd/config.py` (numpy PCG64), which the GPU box can regenerate.

One local shim is needed: `a013_ModelDefinition` imports the training-only
`a008_loss`, which needs kornia (absent); a stub module is registered before the import.
`MyLoss` is never touched by the forward path.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
_stub = types.ModuleType("a008_loss")
_stub.MyLoss = object
sys.modules["a008_loss"] = _stub

from a001_WindowAttention import WindowAttention            # noqa: E402
from a005_BasicBlock import BasicBlock                      # noqa: E402
from a006_PaddingOperation import MyPadding                 # noqa: E402
from a010_StateRecorder import StateRecorder                # noqa: E402
from a011_PatchOperation import PatchMergingAndLinearLayer  # noqa: E402
from a009_NormalAndShiftWinsBlockPair import NormalAndShiftWinsBlockPair  # noqa: E402
from a012_SelfAndCrossBlockPair import SelfAndCrossBlockPair  # noqa: E402
from a013_ModelDefinition import MyModel                    # noqa: E402

from swin_unet_image_fusion_amd.config import (CONFIGS, alias_groups_from_tensors,  # noqa: E402
                                               load_recipe_into, synthetic_pair)

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)


def randn(shape, seed):
    return torch.from_numpy(np.random.default_rng(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32))


def key_table(module):
    sd = module.state_dict()
    alias = alias_groups_from_tensors(sd)
    return {"shapes": {k: list(v.shape) for k, v in sd.items()},
            "alias_of": {k: a for k, a in alias.items() if a != k}}


def save(name, meta, **arrays):
    only = os.environ.get("GOLDEN_ONLY")     # comma-separated fixture names: leave every other file untouched
    if only and name not in only.split(","):
        return
    meta = dict(meta)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.array(json.dumps(meta)),
                        **{k: v.numpy() if isinstance(v, torch.Tensor) else v for k, v in arrays.items()})
    print("wrote", name, {k: tuple(v.shape) for k, v in arrays.items()})


# ---------------------------------------------------------------- WindowAttention (a001)
WA_CASES = [
    # name, C, heads, d, win, (B,H,W), shift, cross, flavor
    ("wa_c8_w4_plain", 8, 2, 4, (4, 4), (2, 8, 12), False, False, "default"),
    ("wa_c8_w4_shift", 8, 2, 4, (4, 4), (2, 8, 12), True, False, "stress"),
    ("wa_c8_w4_shift_cross", 8, 2, 4, (4, 4), (2, 8, 12), True, True, "stress"),
    ("wa_c6_w7_headsne", 6, 4, 3, (7, 7), (1, 14, 7), True, True, "stress"),   # heads*d != C
    ("wa_c24_w8_stage0", 24, 8, 3, (8, 8), (1, 16, 24), True, False, "default"),
    ("wa_c24_w8_cross", 24, 8, 3, (8, 8), (1, 16, 16), False, True, "stress"),
    ("wa_c48_w8_onewin", 48, 8, 6, (8, 8), (2, 8, 8), True, True, "stress"),    # single-window map, mask still applied
    ("wa_c10_w4x8_rect", 10, 2, 5, (4, 8), (1, 8, 16), True, False, "stress"),  # rectangular window
    ("wa_c24_w16", 24, 8, 3, (16, 16), (1, 32, 16), True, True, "stress"),
]


def gen_window_attention():
    for name, c, nh, d, win, (b, h, w), shift, cross, flavor in WA_CASES:
        kw = dict(in_out_dims=c, num_heads=nh, dims_per_head=d, window_size=win, use_cyclic_shift=shift,
                  use_cross_attention=cross, use_qkv_bias=True, attention_drop_ratio=0.0,
                  linear_after_att_drop_ratio=0.0)
        m = WindowAttention(**kw).eval()
        load_recipe_into(m, seed=11, flavor=flavor)
        q = randn((b, c, h, w), 101)
        kv = randn((b, c, h, w), 102) if cross else q
        out = m(q, kv, kv)
        save(name, dict(kind="window_attention", ctor=kw, in_shape=[b, c, h, w], seed_q=101, seed_kv=102,
                        weight_seed=11, flavor=flavor, keys=key_table(m)), expected=out)


# ---------------------------------------------------------------- BasicBlock / pair (a005, a012)
def gen_blocks():
    for name, c, nh, d, win, hid, (b, h, w), shift, cross in [
        ("bb_self_plain", 8, 2, 4, (4, 4), 32, (2, 8, 8), False, False),
        ("bb_self_shift", 8, 2, 4, (4, 4), 32, (2, 8, 12), True, False),
        ("bb_cross_plain", 8, 2, 4, (4, 4), 12, (1, 12, 8), False, True),
        ("bb_cross_shift_w8", 24, 8, 3, (8, 8), 96, (1, 16, 16), True, True),
        ("bb_cross_shift_hid4", 24, 8, 3, (8, 8), 4, (1, 16, 8), True, True),   # decoder last block: hidden 4
    ]:
        kw = dict(in_out_dims=c, num_heads=nh, dims_per_head=d, window_size=win, use_cyclic_shift=shift,
                  use_dual_path=True, use_cross_attr=cross, use_qkv_bias=True, attention_drop_ratio=0.0,
                  linear_after_att_drop_ratio=0.0, mlp_hidden_dims=hid, mlp_activation_func=nn.ELU(inplace=True),
                  mlp_drop_ratio=0.0)
        m = BasicBlock(**kw).eval()
        load_recipe_into(m, seed=12, flavor="stress")
        x, y = randn((b, c, h, w), 201), randn((b, c, h, w), 202)
        ox, oy = m(x.clone(), y.clone())
        kw_meta = {k: v for k, v in kw.items() if k != "mlp_activation_func"}
        save(name, dict(kind="basic_block", ctor=kw_meta, in_shape=[b, c, h, w], seed_x=201, seed_y=202,
                        weight_seed=12, flavor="stress", keys=key_table(m)), expected_x=ox, expected_y=oy)

    # NormalAndShiftWinsBlockPair (a009:90-109): plain-window block then shifted-window block, self or cross
    for name, c, nh, d, win, hid, (b, h, w), cross in [
        ("nswbp_c8_w4_self", 8, 2, 4, (4, 4), 32, (2, 8, 12), False),
        ("nswbp_c24_w8_cross", 24, 8, 3, (8, 8), 96, (1, 16, 24), True),
    ]:
        kw = dict(in_out_dims=c, num_heads=nh, dims_per_head=d, window_size=win, use_dual_path=True, use_cross_attr=cross,
                  use_qkv_bias=True, attention_drop_ratio=0.0, linear_after_att_drop_ratio=0.0,
                  mlp_hidden_dims=hid, mlp_activation_func=nn.ELU(inplace=True), mlp_drop_ratio=0.0)
        m = NormalAndShiftWinsBlockPair(**kw).eval()
        load_recipe_into(m, seed=15, flavor="stress")
        x, y = randn((b, c, h, w), 351), randn((b, c, h, w), 352)
        ox, oy = m(x.clone(), y.clone())
        kw_meta = {k: v for k, v in kw.items() if k != "mlp_activation_func"}
        save(name, dict(kind="normal_and_shift_block_pair", ctor=kw_meta, in_shape=[b, c, h, w], seed_x=351,
                        seed_y=352, weight_seed=15, flavor="stress", keys=key_table(m)),
             expected_x=ox, expected_y=oy)

    for name, c, nh, d, win, hid, (b, h, w) in [
        ("scbp_c8_w4", 8, 2, 4, (4, 4), 32, (2, 8, 8)),
        ("scbp_c24_w8", 24, 8, 3, (8, 8), 96, (1, 16, 24)),
        ("scbp_c12_w7", 12, 4, 3, (7, 7), 24, (1, 14, 14)),
    ]:
        kw = dict(in_out_dims=c, num_heads=nh, dims_per_head=d, window_size=win, use_dual_path=True,
                  use_qkv_bias=True, attention_drop_ratio=0.0, linear_after_att_drop_ratio=0.0,
                  mlp_hidden_dims=hid, mlp_activation_func=nn.ELU(inplace=True), mlp_drop_ratio=0.0)
        m = SelfAndCrossBlockPair(**kw).eval()
        load_recipe_into(m, seed=13, flavor="stress")
        x, y = randn((b, c, h, w), 301), randn((b, c, h, w), 302)
        ox, oy = m(x.clone(), y.clone())
        kw_meta = {k: v for k, v in kw.items() if k != "mlp_activation_func"}
        save(name, dict(kind="self_and_cross_block_pair", ctor=kw_meta, in_shape=[b, c, h, w], seed_x=301,
                        seed_y=302, weight_seed=13, flavor="stress", keys=key_table(m)),
             expected_x=ox, expected_y=oy)


# ---------------------------------------------------------------- patch merge / unmerge (a011), padding (a006)
def gen_patch_and_pad():
    for name, enc, cin, cout, (b, h, w) in [
        ("pm_enc_1_24", True, 1, 24, (2, 16, 12)),
        ("pm_enc_8_16", True, 8, 16, (1, 8, 8)),
        ("pm_dec_16_8", False, 16, 8, (1, 4, 6)),
        ("pm_dec_24_1", False, 24, 1, (2, 8, 8)),
    ]:
        m = PatchMergingAndLinearLayer(belongs_to_encoder=enc, use_dual_path=True, in_dims=cin, out_dims=cout,
                                       patch_merging_size_recorder=StateRecorder(),
                                       merging_or_unmerging_size=(2, 2), activation_func=nn.ELU(inplace=True)).eval()
        load_recipe_into(m, seed=14, flavor="stress")
        x, y = randn((b, cin, h, w), 401), randn((b, cin, h, w), 402)
        ox, oy = m(x.clone(), y.clone())
        save(name, dict(kind="patch_layer", encoder=enc, in_dims=cin, out_dims=cout, in_shape=[b, cin, h, w],
                        seed_x=401, seed_y=402, weight_seed=14, flavor="stress", keys=key_table(m)),
             expected_x=ox, expected_y=oy)

    for name, win, (b, c, h, w) in [("pad_w7_10x13", (7, 7), (1, 3, 10, 13)), ("pad_w2_5x4", (2, 2), (2, 2, 5, 4)),
                                    ("pad_w8_16x16", (8, 8), (1, 2, 16, 16))]:
        fr, pr = StateRecorder(), StateRecorder()
        enc = MyPadding(True, win, True, fr, pr).eval()
        dec = MyPadding(False, win, True, fr, pr).eval()
        x, y = randn((b, c, h, w), 501), randn((b, c, h, w), 502)
        px, py = enc(x, y)
        ux, uy = dec(px, py)
        assert torch.equal(ux, x) and torch.equal(uy, y)
        save(name, dict(kind="padding", window=list(win), in_shape=[b, c, h, w], seed_x=501, seed_y=502),
             padded_x=px, padded_y=py)


# ---------------------------------------------------------------- whole model (a013)
MODEL_CASES = [
    # name, config, (B,H,W), flavor
    ("model_tiny_16", "tiny", (2, 16, 16), "stress"),
    ("model_tiny_pad_18x22", "tiny", (1, 18, 22), "stress"),
    ("model_tiny7_40x36", "tiny7", (1, 40, 36), "stress"),
    ("model_win8_4stage_128", "win8_4stage", (1, 128, 128), "default"),     # BASELINE config 1 (runnable reading)
    ("model_win8_256_default", "win8", (1, 256, 256), "default"),           # BASELINE config 2 shape, B=1
    ("model_win8_256_stress", "win8", (1, 256, 256), "stress"),
    ("model_win7_128", "win7", (1, 128, 128), "default"),                   # 5 stages, win 7 -> padding at depth
    ("model_win7_200", "win7", (1, 200, 200), "default"),                   # a013:427 smoke-loop size
    ("model_win16_512", "win16", (1, 512, 512), "default"),                 # BASELINE config 5 window
    ("model_win8_b2_128x192", "win8_4stage", (2, 128, 192), "stress"),      # deepest map 8x12 -> pad to 8x16
    ("model_win8_512_default", "win8", (1, 512, 512), "default"),           # BASELINE config 3 shape, B=1
    ("model_win16_1024_default", "win16", (1, 1024, 1024), "default"),      # BASELINE config 5 shape, B=1
    ("model_win7_224_default", "win7", (1, 224, 224), "default"),           # the reference's default window (A000:55), every map a multiple of 7
    # the initialisation the reference trains from (a016:42, a016:382-390): kaiming_normal_ weights, zero biases
    ("model_win8_256_kaiming", "win8", (1, 256, 256), "kaiming"),
    ("model_win7_224_kaiming", "win7", (1, 224, 224), "kaiming"),
    ("model_win8_4stage_128_kaiming", "win8_4stage", (2, 128, 128), "kaiming"),
]


def gen_models():
    tables_done = set()
    only = os.environ.get("GOLDEN_ONLY")
    for name, cfg_name, (b, h, w), flavor in MODEL_CASES:
        if only and name not in only.split(","):
            continue
        cfg = CONFIGS[cfg_name]
        m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
        load_recipe_into(m, seed=0, flavor=flavor)
        if cfg_name not in tables_done:
            with open(os.path.join(OUT, f"state_keys_{cfg_name}.json"), "w") as f:
                json.dump(key_table(m), f)
            tables_done.add(cfg_name)
        ir, vis = synthetic_pair(b, h, w)
        out = m(torch.from_numpy(ir), torch.from_numpy(vis))
        save(name, dict(kind="model", config=cfg_name, in_shape=[b, 1, h, w], seed_ir=1, seed_vis=2,
                        weight_seed=0, flavor=flavor, keys_file=f"state_keys_{cfg_name}.json"), expected=out)

    # the reference raises on BASELINE config 1 as literally written: record the class
    cfg = CONFIGS["win8"]
    m = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval()
    ir, vis = synthetic_pair(1, 128, 128)
    try:
        m(torch.from_numpy(ir), torch.from_numpy(vis))
        err = None
    except Exception as e:  # noqa: BLE001
        err = type(e).__name__
    with open(os.path.join(OUT, "config1_literal_error.json"), "w") as f:
        json.dump({"config": "win8", "in_shape": [1, 1, 128, 128], "raises": err}, f)
    print("config-1 literal raises:", err)


if __name__ == "__main__":
    which = sys.argv[1:] or ["wa", "blocks", "patch", "models"]
    if "wa" in which:
        gen_window_attention()
    if "blocks" in which:
        gen_blocks()
    if "patch" in which:
        gen_patch_and_pad()
    if "models" in which:
        gen_models()
