"""CPU ORACLE — test infrastructure only, never the product path.

A plain-PyTorch (CPU, fp32) restatement of the forward path of
RainbowZL0/swin-unet-image-fusion, written from the algorithm, not
from the reference's code structure: purely functional, weights come in as a flat dict that
uses the reference's state_dict key names, window logic is index arithmetic instead of
einops patterns, the shift mask is computed from coordinates.

Who may import this file: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` — as the checker / the reported CPU baseline.  The product path
(`swin_unet_image_fusion_amd`) never imports it and fails loudly without its HIP library.

Parity status: PINNED.  `oracle/make_golden.py` imports the real reference on CPU in the
build container and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every
function below against those vectors (<=1e-5 relative).  The reference itself has no
assertion-bearing tests or golden vectors.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------
# primitive pieces
# ----------------------------------------------------------------------------------------
def layer_norm_channels(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNorm over C of an NCHW map (a004_AddAndLayerNormWithOtherModule.py:54-72:
    permute to NHWC, nn.LayerNorm(C), permute back)."""
    xt = x.permute(0, 2, 3, 1)
    xt = F.layer_norm(xt, (xt.shape[-1],), weight, bias, eps)
    return xt.permute(0, 3, 1, 2)


def relative_position_bias(table: Tensor, win: Tuple[int, int]) -> Tensor:
    """(t, t) bias, bias[i, j] = table[yj - yi + wh - 1, xj - xi + ww - 1] with i the query
    token and j the key token, tokens row-major inside the window
    (a001_WindowAttention.py:100-125 builds the indices, :127-144 gathers)."""
    wh, ww = win
    ys = torch.arange(wh).repeat_interleave(ww)
    xs = torch.arange(ww).repeat(wh)
    dy = ys[None, :] - ys[:, None] + (wh - 1)
    dx = xs[None, :] - xs[:, None] + (ww - 1)
    return table[dy, dx]


def shift_region_ids(h: int, w: int, win: Tuple[int, int]) -> Tensor:
    """(h, w) int map of the 3x3 region labels used by the cyclic-shift mask
    (a001:217-247): rows [0,h-wh) -> 0, [h-wh, h-wh//2) -> 1, [h-wh//2, h) -> 2, same for
    columns, label = 3*row_band + col_band.  Labels live on the *shifted* map."""
    wh, ww = win
    yy = torch.arange(h)
    xx = torch.arange(w)
    band_y = (yy >= h - wh).int() + (yy >= h - wh // 2).int()
    band_x = (xx >= w - ww).int() + (xx >= w - ww // 2).int()
    return band_y[:, None] * 3 + band_x[None, :]


def window_partition(x: Tensor, win: Tuple[int, int]) -> Tensor:
    """(B, C, H, W) -> (B*nWy*nWx, wh*ww, C); windows ordered (b, wy, wx), tokens row-major
    (a001:154-172)."""
    b, c, h, w = x.shape
    wh, ww = win
    x = x.reshape(b, c, h // wh, wh, w // ww, ww)
    return x.permute(0, 2, 4, 3, 5, 1).reshape(-1, wh * ww, c)


def window_reverse(t: Tensor, win: Tuple[int, int], b: int, h: int, w: int) -> Tensor:
    """inverse of window_partition (a001:373-398)."""
    wh, ww = win
    c = t.shape[-1]
    t = t.reshape(b, h // wh, w // ww, wh, ww, c)
    return t.permute(0, 5, 1, 3, 2, 4).reshape(b, c, h, w)


def window_attention(sd: State, prefix: str, q: Tensor, k: Tensor, v: Tensor, *,
                     num_heads: int, dims_per_head: int, window_size: Tuple[int, int],
                     use_cyclic_shift: bool) -> Tensor:
    """WindowAttention.forward(q, k, v) (a001:448-474), SURVEY.md §3.2 steps 1-11.
    `prefix` ends with '.', e.g. '...auto_path_win_att.window_attention_x.'."""
    b, c, h, w = q.shape
    wh, ww = window_size
    if h % wh or w % ww:
        raise ValueError(f"map {h}x{w} is not a multiple of the window {wh}x{ww}")
    sh, sw = wh // 2, ww // 2
    if use_cyclic_shift:  # a001:419-446: roll by (-wh//2, -ww//2)
        q, k, v = (torch.roll(t, shifts=(-sh, -sw), dims=(2, 3)) for t in (q, k, v))
    t = wh * ww
    qw, kw, vw = (window_partition(z, window_size) for z in (q, k, v))
    lin = lambda z, name: F.linear(z, sd[prefix + name + ".weight"], sd.get(prefix + name + ".bias"))
    # a001:196-215 — three Linear layers, then heads split as channel = head*d + j (a001:174-194)
    split = lambda z: z.reshape(z.shape[0], t, num_heads, dims_per_head).permute(0, 2, 1, 3)
    qh, kh, vh = split(lin(qw, "q_for_heads")), split(lin(kw, "k_for_heads")), split(lin(vw, "v_for_heads"))
    scores = torch.matmul(qh, kh.transpose(-1, -2)) * (dims_per_head ** -0.5)  # a001:333-335
    scores = scores + relative_position_bias(sd[prefix + "relative_position_bias_table"], window_size)
    if use_cyclic_shift:  # a001:274-315: scores[mask] = -1e10, mask broadcast over batch & heads
        ids = shift_region_ids(h, w, window_size)[None, None].float()
        ids = window_partition(ids, window_size)[..., 0]              # (nW, t)
        mask = ids[:, :, None] != ids[:, None, :]                     # (nW, t, t)
        n_win = mask.shape[0]
        scores = scores.reshape(b, n_win, num_heads, t, t)
        scores = scores.masked_fill(mask[None, :, None], -1e10)
        scores = scores.reshape(b * n_win, num_heads, t, t)
    probs = torch.softmax(scores, dim=-1)                             # a001:349
    out = torch.matmul(probs, vh)                                     # a001:353
    out = out.permute(0, 2, 1, 3).reshape(-1, t, num_heads * dims_per_head)  # a001:357-371
    out = F.linear(out, sd[prefix + "linear_projection.weight"], sd[prefix + "linear_projection.bias"])
    out = window_reverse(out, window_size, b, h, w)                   # a001:400-417
    if use_cyclic_shift:
        out = torch.roll(out, shifts=(sh, sw), dims=(2, 3))           # a001:471-473
    return out


def auto_path_win_att(sd: State, prefix: str, x: Tensor, y: Tensor, *, cross: bool, **kw) -> Tuple[Tensor, Tensor]:
    """AutoPathWinAtt.forward (a002_AutoPathWinAtt.py:58-82), dual path."""
    px, py = prefix + "window_attention_x.", prefix + "window_attention_y."
    if cross:
        return window_attention(sd, px, x, y, y, **kw), window_attention(sd, py, y, x, x, **kw)
    return window_attention(sd, px, x, x, x, **kw), window_attention(sd, py, y, y, y, **kw)


def auto_path_mlp(sd: State, prefix: str, x: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
    """AutoPathMLP.forward (a003_AutoPathMLP.py:21-50): 1x1 conv, ELU, 1x1 conv per stream."""
    def one(z, s):
        z = F.conv2d(z, sd[f"{prefix}mlp_{s}_1.weight"], sd[f"{prefix}mlp_{s}_1.bias"])
        z = F.elu(z)
        return F.conv2d(z, sd[f"{prefix}mlp_{s}_2.weight"], sd[f"{prefix}mlp_{s}_2.bias"])
    return one(x, "x"), one(y, "y")


def basic_block(sd: State, prefix: str, x: Tensor, y: Tensor, *, cross: bool, shift: bool,
                num_heads: int, dims_per_head: int, window_size: Tuple[int, int]) -> Tuple[Tensor, Tensor]:
    """BasicBlock.forward (a005_BasicBlock.py:127-145): stage_1 = pre-LN attention + residual,
    stage_2 = pre-LN MLP + residual (a004:29-38)."""
    if cross and bool((x == y).all()):
        # a005:111-118 calls exit(); the oracle raises instead (documented deviation)
        raise ValueError("cross attention received identical x and y")
    nx = layer_norm_channels(x, sd[prefix + "stage_1.norm_layer_1.weight"], sd[prefix + "stage_1.norm_layer_1.bias"])
    ny = layer_norm_channels(y, sd[prefix + "stage_1.norm_layer_2.weight"], sd[prefix + "stage_1.norm_layer_2.bias"])
    ax, ay = auto_path_win_att(sd, prefix + "auto_path_win_att.", nx, ny, cross=cross, num_heads=num_heads,
                               dims_per_head=dims_per_head, window_size=window_size, use_cyclic_shift=shift)
    x, y = x + ax, y + ay
    nx = layer_norm_channels(x, sd[prefix + "stage_2.norm_layer_1.weight"], sd[prefix + "stage_2.norm_layer_1.bias"])
    ny = layer_norm_channels(y, sd[prefix + "stage_2.norm_layer_2.weight"], sd[prefix + "stage_2.norm_layer_2.bias"])
    mx, my = auto_path_mlp(sd, prefix + "auto_path_mlp.", nx, ny)
    return x + mx, y + my


def normal_and_shift_block_pair(sd: State, prefix: str, x: Tensor, y: Tensor, *, cross: bool, **kw) -> Tuple[Tensor, Tensor]:
    """NormalAndShiftWinsBlockPair.forward (a009_NormalAndShiftWinsBlockPair.py:90-109): the plain-window
    BasicBlock, then the shifted-window BasicBlock (a009:102-105)."""
    for blk, shift in (("normal_window_block.", False), ("shifted_window_block.", True)):
        x, y = basic_block(sd, prefix + blk, x, y, cross=cross, shift=shift, **kw)
    return x, y


def self_and_cross_block_pair(sd: State, prefix: str, x: Tensor, y: Tensor, **kw) -> Tuple[Tensor, Tensor]:
    """SelfAndCrossBlockPair.forward (a012:70-78) = self pair then cross pair, each pair =
    normal-window block then shifted-window block (a009:90-109)."""
    for group, cross in (("self_att_block.", False), ("cross_att_block.", True)):
        x, y = normal_and_shift_block_pair(sd, prefix + group, x, y, cross=cross, **kw)
    return x, y


def pad_to_multiple(x: Tensor, win: Tuple[int, int]) -> Tuple[Tensor, Tuple[int, int]]:
    """MyPadding encoder side (a006_PaddingOperation.py:54-56, 122-131): reflect-pad bottom /
    right up to the next multiple of `win`.  torch raises RuntimeError when pad >= dim."""
    h, w = x.shape[-2:]
    ph = (win[0] - h % win[0]) % win[0]
    pw = (win[1] - w % win[1]) % win[1]
    if ph == 0 and pw == 0:
        return x, (0, 0)
    return F.pad(x, (0, pw, 0, ph), mode="reflect"), (ph, pw)


def crop_padding(x: Tensor, pad: Tuple[int, int]) -> Tensor:
    """MyPadding decoder side (a006:133-146)."""
    h, w = x.shape[-2:]
    return x[:, :, : h - pad[0], : w - pad[1]]


def space_to_depth(x: Tensor, m: Tuple[int, int]) -> Tensor:
    """a011_PatchOperation.py:73-94: out channel = (ph*mw + pw)*C + c."""
    b, c, h, w = x.shape
    mh, mw = m
    x = x.reshape(b, c, h // mh, mh, w // mw, mw)
    return x.permute(0, 3, 5, 1, 2, 4).reshape(b, mh * mw * c, h // mh, w // mw)


def depth_to_space(x: Tensor, m: Tuple[int, int]) -> Tensor:
    """a011:96-117 (and the reshape/permute twin a011:119-145)."""
    b, cc, h, w = x.shape
    mh, mw = m
    c = cc // (mh * mw)
    x = x.reshape(b, mh, mw, c, h, w)
    return x.permute(0, 3, 4, 1, 5, 2).reshape(b, c, h * mh, w * mw)


def patch_layer(sd: State, prefix: str, x: Tensor, y: Tensor, *, encoder: bool,
                merging_size: Tuple[int, int]) -> Tuple[Tensor, Tensor]:
    """PatchMergingAndLinearLayer.forward (a011:244-264).  Encoder order: merge, 1x1 conv, LN,
    ELU (a011:236-239); decoder order: 1x1 conv, LN (over 4*Cout), unmerge, ELU (a011:241)."""
    outs = []
    for z, s in ((x, "x"), (y, "y")):
        wgt, bias = sd[f"{prefix}mlp_layer_{s}.weight"], sd[f"{prefix}mlp_layer_{s}.bias"]
        g, bt = sd[f"{prefix}layer_norm_{s}.weight"], sd[f"{prefix}layer_norm_{s}.bias"]
        if encoder:
            z = space_to_depth(z, merging_size)
            z = layer_norm_channels(F.conv2d(z, wgt, bias), g, bt)
        else:
            z = layer_norm_channels(F.conv2d(z, wgt, bias), g, bt)
            z = depth_to_space(z, merging_size)
        outs.append(F.elu(z))
    return outs[0], outs[1]


def final_head(sd: State, x: Tensor, y: Tensor, ksize: int = 3, training: bool = False) -> Tensor:
    """MyModel.do_final_layer (a013_ModelDefinition.py:126-152): cat -> conv kxk (reflect
    'same') -> BatchNorm2d -> ELU -> conv kxk (reflect 'same').  BatchNorm2d as nn.BatchNorm2d
    runs it: running statistics under eval() (the inference path), batch statistics plus an
    in-place running-statistics update (momentum 0.1) under train() (a016_train.py:137)."""
    p = ksize // 2
    z = torch.cat([x, y], dim=1)
    z = F.conv2d(F.pad(z, (p, p, p, p), mode="reflect"), sd["final_layer.0.weight"], sd["final_layer.0.bias"])
    z = F.batch_norm(z, sd["final_layer.1.running_mean"], sd["final_layer.1.running_var"],
                     sd["final_layer.1.weight"], sd["final_layer.1.bias"], training=training, momentum=0.1, eps=1e-5)
    z = F.elu(z)
    return F.conv2d(F.pad(z, (p, p, p, p), mode="reflect"), sd["final_layer.3.weight"], sd["final_layer.3.bias"])


# ----------------------------------------------------------------------------------------
# whole model
# ----------------------------------------------------------------------------------------
def model_forward(sd: State, cfg, in_x: Tensor, in_y: Tensor, training: bool = False) -> Tensor:
    """MyModel.forward (a013:209-230).  `cfg` is a FusionConfig-like object (window_size,
    merging_size, in_dims_list, out_dims_list, att_num_heads, att_dims_per_head_ratio,
    final_conv_layer_kernel_size).  Stage module order: encoder [pad2, merge, padW, blocks],
    decoder reversed (a013:236-314) so decoder keys are .0 blocks, .2 merge."""
    win, msz = tuple(cfg.window_size), tuple(cfg.merging_size)
    n = len(cfg.in_dims_list)
    x, y = in_x, in_y
    pads: List[Tuple[int, int]] = []      # the shared LIFO of a006 (a013:56-58)
    skips: List[Tuple[Tensor, Tensor]] = []
    for s in range(n):
        kw = dict(num_heads=cfg.att_num_heads,
                  dims_per_head=math.floor(cfg.out_dims_list[s] * cfg.att_dims_per_head_ratio),
                  window_size=win)
        x, p = pad_to_multiple(x, msz); y, _ = pad_to_multiple(y, msz); pads.append(p)
        x, y = patch_layer(sd, f"encoder_list.{s}.1.", x, y, encoder=True, merging_size=msz)
        x, p = pad_to_multiple(x, win); y, _ = pad_to_multiple(y, win); pads.append(p)
        x, y = self_and_cross_block_pair(sd, f"encoder_list.{s}.3.", x, y, **kw)
        if s < n - 1:
            skips.append((x, y))
    for j in range(n):
        lvl = n - 1 - j
        kw = dict(num_heads=cfg.att_num_heads,
                  dims_per_head=math.floor(cfg.out_dims_list[lvl] * cfg.att_dims_per_head_ratio),
                  window_size=win)
        if j > 0:
            hx, hy = skips.pop()
            x, y = x + hx, y + hy
        x, y = self_and_cross_block_pair(sd, f"decoder_list.{j}.0.", x, y, **kw)
        p = pads.pop(); x, y = crop_padding(x, p), crop_padding(y, p)
        x, y = patch_layer(sd, f"decoder_list.{j}.2.", x, y, encoder=False, merging_size=msz)
        p = pads.pop(); x, y = crop_padding(x, p), crop_padding(y, p)
    return final_head(sd, x, y, cfg.final_conv_layer_kernel_size, training=training)
