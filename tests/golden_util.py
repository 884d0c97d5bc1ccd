"""Load golden fixtures (tests/golden, written by oracle/make_golden.py from the real
reference) and regenerate their inputs / weights from the deterministic recipes."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict

import numpy as np
import torch

from swin_unet_image_fusion_amd.config import CONFIGS, make_state_arrays, synthetic_pair

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases(kind: str):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        with np.load(p) as z:
            meta = json.loads(str(z["meta"]))
        if meta["kind"] == kind:
            out.append(os.path.basename(p)[:-4])
    return out


def load(name: str):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrays


def key_table(meta) -> dict:
    if "keys" in meta:
        return meta["keys"]
    with open(os.path.join(GOLDEN, meta["keys_file"])) as f:
        return json.load(f)


def recipe_state(meta) -> Dict[str, torch.Tensor]:
    kt = key_table(meta)
    shapes = {k: tuple(v) for k, v in kt["shapes"].items()}
    arrays = make_state_arrays(shapes, kt["alias_of"], seed=meta["weight_seed"], flavor=meta["flavor"])
    return {k: torch.from_numpy(v) for k, v in arrays.items()}


def randn(shape, seed) -> torch.Tensor:
    return torch.from_numpy(np.random.default_rng(np.random.PCG64(seed)).standard_normal(tuple(shape)).astype(np.float32))


def model_inputs(meta):
    b, _, h, w = meta["in_shape"]
    ir, vis = synthetic_pair(b, h, w, meta["seed_ir"], meta["seed_vis"])
    return torch.from_numpy(ir), torch.from_numpy(vis)


def rel_err(a: torch.Tensor, ref: torch.Tensor):
    """(rel-L2, max|err|/max|ref|) — the two parity metrics of SURVEY.md §7 hard part 3."""
    a, ref = a.double(), ref.double()
    l2 = float((a - ref).norm() / ref.norm().clamp_min(1e-30))
    mx = float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    return l2, mx
