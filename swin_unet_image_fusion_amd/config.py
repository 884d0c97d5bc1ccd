"""Model hyper-parameters and the deterministic weight recipe.

The reference keeps its hyper-parameters as module constants (A000_CONFIG.py:55-69) and
passes them as `MyModel(...)` kwargs (a013_ModelDefinition.py:18-38).  Nothing here touches
the GPU or the oracle; it is pure host logic shared by the product path, the tests and
`bench.py`.

Weight recipe: no checkpoint ships with the reference, so every parity
fixture and every benchmark uses weights drawn by a framework-independent recipe: numpy
PCG64 with a fixed seed, iterating the *unique* tensors of a state_dict in sorted-name
order (an alias group — the reference registers several sub-modules under two names,
a005_BasicBlock.py:51-82, a003_AutoPathMLP.py:25-31 — is represented by its
lexicographically smallest key).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Tuple

import numpy as np


@dataclass(frozen=True)
class FusionConfig:
    """Constructor arguments of the reference `MyModel` that shape the forward path."""

    window_size: Tuple[int, int] = (8, 8)
    merging_size: Tuple[int, int] = (2, 2)
    in_dims_list: Tuple[int, ...] = (1, 24, 48, 96, 192)
    out_dims_list: Tuple[int, ...] = (24, 48, 96, 192, 384)
    att_num_heads: int = 8
    att_dims_per_head_ratio: float = 1 / 8
    mlp_hidden_dims_ratio: int = 4
    final_conv_layer_kernel_size: int = 3

    @property
    def n_levels(self) -> int:
        return len(self.in_dims_list)

    def dims_per_head(self, level: int) -> int:
        # a013_ModelDefinition.py:174,191 — both encoder and decoder use out_dims_list[j]
        return math.floor(self.out_dims_list[level] * self.att_dims_per_head_ratio)

    def model_kwargs(self, activation) -> dict:
        """kwargs accepted by `MyModel(...)` (reference a013:18-38 and this package)."""
        return dict(
            window_size=tuple(self.window_size),
            merging_size=tuple(self.merging_size),
            in_dims_list=list(self.in_dims_list),
            out_dims_list=list(self.out_dims_list),
            att_num_heads=self.att_num_heads,
            att_dims_per_head_ratio=self.att_dims_per_head_ratio,
            attention_drop_ratio=0.0,
            linear_after_att_drop_ratio=0.0,
            mlp_hidden_dims_ratio=self.mlp_hidden_dims_ratio,
            mlp_activation_func=activation,
            mlp_drop_ratio=0.0,
            final_layer_att_dims_per_head_ratio=self.att_dims_per_head_ratio,
            final_conv_layer_kernel_size=self.final_conv_layer_kernel_size,
            final_layer_mlp_hidden_dims_ratio=self.mlp_hidden_dims_ratio,
        )


# Named configurations used by tests / bench.
CONFIGS: Dict[str, FusionConfig] = {
    # A000_CONFIG.py:55-69 with window 8 — BASELINE.json configs 2-4
    "win8": FusionConfig(),
    # reference default window (A000:55)
    "win7": FusionConfig(window_size=(7, 7)),
    # BASELINE.json config 5
    "win16": FusionConfig(window_size=(16, 16)),
    # BASELINE.json config 1, runnable reading (4 stages at 128x128, win 8)
    "win8_4stage": FusionConfig(in_dims_list=(1, 24, 48, 96), out_dims_list=(24, 48, 96, 192)),
    # tiny net for fast per-op / whole-model fixtures
    "tiny": FusionConfig(window_size=(4, 4), in_dims_list=(1, 8), out_dims_list=(8, 16),
                         att_num_heads=2, att_dims_per_head_ratio=1 / 2),
    "tiny7": FusionConfig(window_size=(7, 7), in_dims_list=(1, 8, 16), out_dims_list=(8, 16, 24),
                          att_num_heads=4, att_dims_per_head_ratio=1 / 4),
}


def _canonical_groups(shapes_by_key: Dict[str, Tuple[int, ...]],
                      alias_of: Dict[str, str]) -> List[str]:
    return sorted({alias_of.get(k, k) for k in shapes_by_key})


def alias_groups_from_tensors(state_dict) -> Dict[str, str]:
    """Map every key of a torch state_dict to the smallest key sharing its storage."""
    by_ptr: Dict[Tuple[int, Tuple[int, ...]], str] = {}
    for k in sorted(state_dict.keys()):
        t = state_dict[k]
        ident = (t.data_ptr(), tuple(t.shape), str(t.dtype))
        by_ptr.setdefault(ident, k)
    out = {}
    for k, t in state_dict.items():
        out[k] = by_ptr[(t.data_ptr(), tuple(t.shape), str(t.dtype))]
    return out


def _draw(rng: np.random.Generator, key: str, shape: Tuple[int, ...], flavor: str) -> np.ndarray:
    """One tensor of the recipe.  `default` follows the reference constructors' distributions
    (nn.Linear / nn.Conv2d U(+-1/sqrt(fan_in)), LayerNorm 1/0, bias table N(0,1) a001:76,
    BatchNorm 1/0/0/1).  `stress` perturbs every affine / statistic so that a kernel that
    drops one of them cannot pass parity.  `kaiming` is what the reference TRAINS from
    (a016_train.py:42 applies init_params, a016:382-390: `init.kaiming_normal_` = N(0, 2/fan_in) on every
    Linear / Conv2d weight, zero biases; norms and bias tables keep their constructor values)."""
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "buffer_to_show_device":
        return np.zeros(shape, dtype=np.float32)
    if leaf == "relative_position_bias_table":
        return rng.standard_normal(shape).astype(np.float32)
    is_norm = ("norm_layer_" in key) or ("layer_norm_" in key) or key.startswith("final_layer.1.")
    if is_norm:
        if flavor in ("default", "kaiming"):
            if leaf in ("weight", "running_var"):
                return np.ones(shape, dtype=np.float32)
            return np.zeros(shape, dtype=np.float32)
        if leaf == "weight":
            return rng.uniform(0.5, 1.5, shape).astype(np.float32)
        if leaf == "running_var":
            return rng.uniform(0.5, 2.0, shape).astype(np.float32)
        return rng.uniform(-0.5, 0.5, shape).astype(np.float32)  # bias / running_mean
    # Linear / Conv weight or bias: fan_in from the weight shape; bias uses its out size's
    # layer fan_in which we do not know here, so the caller passes fan_in via closure.
    raise KeyError(key)


def make_state_arrays(shapes_by_key: Dict[str, Tuple[int, ...]], alias_of: Dict[str, str],
                      seed: int = 0, flavor: str = "default") -> Dict[str, np.ndarray]:
    """Draw every unique tensor; returns arrays for *all* keys (aliases share the array)."""
    assert flavor in ("default", "stress", "kaiming")
    rng = np.random.default_rng(np.random.PCG64(seed))
    canon = _canonical_groups(shapes_by_key, alias_of)
    drawn: Dict[str, np.ndarray] = {}
    # fan_in of a Linear/Conv layer comes from its weight; look it up for the bias.
    def fan_in_for(key: str) -> int:
        wkey = key.rsplit(".", 1)[0] + ".weight"
        wshape = shapes_by_key[wkey]
        return int(np.prod(wshape[1:]))
    gain = 1.0 if flavor == "default" else 1.7
    for key in canon:
        shape = tuple(shapes_by_key[key])
        try:
            drawn[key] = _draw(rng, key, shape, flavor)
        except KeyError:
            if flavor == "kaiming":   # a016:382-390
                leaf = key.rsplit(".", 1)[-1]
                if leaf == "bias":
                    drawn[key] = np.zeros(shape, dtype=np.float32)
                else:
                    drawn[key] = (rng.standard_normal(shape) * math.sqrt(2.0 / fan_in_for(key))).astype(np.float32)
                continue
            bound = gain / math.sqrt(fan_in_for(key))
            drawn[key] = rng.uniform(-bound, bound, shape).astype(np.float32)
    return {k: drawn[alias_of.get(k, k)] for k in shapes_by_key}


def load_recipe_into(module, seed: int = 0, flavor: str = "default") -> None:
    """Fill a torch module (the reference's or this package's) in place with the recipe."""
    import torch

    sd = module.state_dict()
    alias_of = alias_groups_from_tensors(sd)
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    arrays = make_state_arrays(shapes, alias_of, seed=seed, flavor=flavor)
    with torch.no_grad():
        for k in sorted(set(alias_of.values())):
            sd[k].copy_(torch.from_numpy(arrays[k]).to(sd[k].dtype))


def synthetic_pair(batch: int, height: int, width: int, seed_ir: int = 1, seed_vis: int = 2):
    """IR / visible inputs ~U[0,1) fp32 (B,1,H,W); independent streams (a005:111-118 forbids
    identical x and y in cross blocks; a015:57-60 scales images to [0,1])."""
    ir = np.random.default_rng(np.random.PCG64(seed_ir)).random((batch, 1, height, width), dtype=np.float32)
    vis = np.random.default_rng(np.random.PCG64(seed_vis)).random((batch, 1, height, width), dtype=np.float32)
    return ir, vis
