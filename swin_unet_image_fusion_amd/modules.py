"""Host-side mirror of the reference's module API, backed by libswinfuse (HIP, gfx950).

Same class names, constructor signatures, `forward` / `forward_` signatures and state_dict
keys as the reference (a001..a013), so `load_state_dict` of a reference checkpoint is strict-
compatible and callers (a016:150, a017:72) can switch by changing an import.  The classes own
ordinary `nn.Parameter`s (inside nn.Linear / nn.Conv2d / nn.LayerNorm / nn.BatchNorm2d
containers, which are used purely as named parameter holders); every `forward` marshals raw
device pointers into the C-ABI.  No arithmetic of the hot path runs in PyTorch, and there is
no CPU fallback: without the built library, or on a CPU tensor, forward raises.

Only the eval()/no_grad forward exists (SURVEY.md §8b "Mode").
"""
from __future__ import annotations

import ctypes as C
import math
from collections import deque
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _lib as L

__all__ = ["WindowAttention", "AutoPathWinAtt", "AutoPathMLP", "AddAndLayerNormWithOtherModule", "BasicBlock",
           "NormalAndShiftWinsBlockPair", "SelfAndCrossBlockPair", "PatchMergingAndLinearLayer", "MyPadding",
           "StateRecorder", "MyModel", "get_encoder_or_decoder_block"]


# ----------------------------------------------------------------------------------------------
# plumbing: pointers, streams, workspaces, layout changes
# ----------------------------------------------------------------------------------------------
def _ptr(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise RuntimeError("libswinfuse needs contiguous fp32 tensors on the GPU "
                           f"(got dtype={t.dtype}, device={t.device}, contiguous={t.is_contiguous()})")
    return t.data_ptr()


_BOUND_DEVICE: Optional[int] = None


def _stream(device) -> int:
    """Current HIP stream of `device`.  libswinfuse launches on the calling thread's current HIP device and caches
    per-device facts (CU count, kernel attributes) once per process, so a process is bound to ONE GPU — the deployment
    model of this package (one process per GPU, shard.py).  A second device, or a current device that differs from the
    tensors' device, raises instead of launching on the wrong GPU."""
    global _BOUND_DEVICE
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if _BOUND_DEVICE is None:
        _BOUND_DEVICE = idx
    if idx != _BOUND_DEVICE:
        raise RuntimeError(f"libswinfuse is bound to cuda:{_BOUND_DEVICE} in this process (one process per GPU); "
                           f"got a tensor on cuda:{idx}")
    if torch.cuda.current_device() != idx:
        raise RuntimeError(f"tensors live on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device() first (the library launches on the current HIP device)")
    return torch.cuda.current_stream(device).cuda_stream


_WS = {}


def _workspace(nbytes: int, device) -> Tuple[Optional[int], int]:
    """Grow-only scratch buffer per (device, stream); the library only borrows it for one call."""
    if nbytes <= 0:
        return None, 0
    key = (torch.device(device).index, _stream(device))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf.data_ptr(), buf.numel()


def _workspace_tensor(device, stream_handle: int) -> Optional[Tensor]:
    """The scratch buffer currently registered for (device, raw stream handle), or None.  Whoever bakes its address into a
    captured hipGraph (shard.ShardedFusion) keeps this tensor alive: _workspace() replaces the registered buffer when a later
    call on the same stream handle needs more bytes, and torch recycles stream handles from a small pool."""
    return _WS.get((torch.device(device).index, stream_handle))


def _check_forward_only(module: nn.Module, *tensors: Optional[Tensor]) -> None:
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError("libswinfuse provides the forward pass only; call it under torch.no_grad() "
                           "with inputs that do not require grad")
    for t in tensors:
        if t is not None and (t.dim() != 4):
            raise ValueError(f"expected a 4-D (batch, channels, height, width) tensor, got shape {tuple(t.shape)}")


def _wants_grad(module: nn.Module, *tensors: Optional[Tensor]) -> bool:
    """True when torch.autograd is recording and an input or a parameter of `module` takes part in it (training side)."""
    return torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors) or
                                        any(p.requires_grad for p in module.parameters()))


def _to_nhwc(t: Tensor) -> Tensor:
    b, c, h, w = t.shape
    t = t.contiguous()
    if c == 1:
        return t.view(b, h, w, 1)
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=t.device)
    L.check(L.lib().swf_nchw_to_nhwc(_ptr(t), _ptr(out), b, c, h, w, _stream(t.device)))
    return out


def _to_nchw(t: Tensor) -> Tensor:
    b, h, w, c = t.shape
    if c == 1:
        return t.view(b, 1, h, w)
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=t.device)
    L.check(L.lib().swf_nhwc_to_nchw(_ptr(t), _ptr(out), b, c, h, w, _stream(t.device)))
    return out


def _lin(mod: nn.Module) -> L.Linear:
    return L.Linear(_ptr(mod.weight), _ptr(mod.bias) if mod.bias is not None else None)


def _norm(mod: nn.LayerNorm) -> L.Norm:
    return L.Norm(_ptr(mod.weight), _ptr(mod.bias))


def _require_elu(act: nn.Module) -> None:
    if not isinstance(act, nn.ELU) or float(act.alpha) != 1.0:
        raise NotImplementedError("the HIP kernels implement the reference's configured activation, "
                                  f"nn.ELU(alpha=1) (A000_CONFIG.py:64); got {act!r}")


def _precision_code(p) -> int:
    if p in (L.PREC_FP32, "fp32"):
        return L.PREC_FP32
    if p in (L.PREC_FAST, "fast"):
        return L.PREC_FAST
    raise ValueError(f"precision must be 'fast' or 'fp32', got {p!r}")


class StateRecorder:
    """LIFO used for pad sizes, shapes and U-Net skips (reference a010_StateRecorder.py:1-18)."""

    def __init__(self):
        self.record_stack = []

    def record(self, new_item):
        self.record_stack.append(new_item)

    def read(self):
        return self.record_stack.pop()

    def delete_all(self):
        self.record_stack.clear()

    def peek(self):
        return self.record_stack[-1] if self.record_stack else None


class _FwdAlias:
    def forward_(self, *a, **kw):  # every reference module has this alias (a001:476, a012:80, a013:232)
        return self(*a, **kw)


# ----------------------------------------------------------------------------------------------
# a001 WindowAttention
# ----------------------------------------------------------------------------------------------
class WindowAttention(_FwdAlias, nn.Module):
    """Drop-in for a001_WindowAttention.WindowAttention (ctor a001:9-20, forward a001:448-474)."""

    def __init__(self, in_out_dims: int, num_heads: int, dims_per_head: int, window_size: tuple,
                 use_cyclic_shift: bool, use_cross_attention: bool, use_qkv_bias: bool,
                 attention_drop_ratio: float, linear_after_att_drop_ratio: float):
        super().__init__()
        self.in_out_dims, self.num_heads, self.dims_per_head = in_out_dims, num_heads, dims_per_head
        self.window_size = tuple(window_size)
        self.use_cyclic_shift, self.use_cross_attention, self.use_qkv_bias = use_cyclic_shift, use_cross_attention, use_qkv_bias
        self.attention_drop_ratio, self.linear_after_att_drop_ratio = attention_drop_ratio, linear_after_att_drop_ratio
        self.qk_scale = dims_per_head ** -0.5
        self.feature_shape_hw: tuple = tuple()
        self.precision = "fast"   # arithmetic tier of this module's own forward: "fast" (MFMA, <= 1e-3) or "fp32" (exact)
        hd = num_heads * dims_per_head
        self.q_for_heads = nn.Linear(in_out_dims, hd, bias=use_qkv_bias)
        self.k_for_heads = nn.Linear(in_out_dims, hd, bias=use_qkv_bias)
        self.v_for_heads = nn.Linear(in_out_dims, hd, bias=use_qkv_bias)
        self.linear_projection = nn.Linear(hd, in_out_dims)
        self.relative_position_bias_table = nn.Parameter(
            torch.randn(2 * self.window_size[0] - 1, 2 * self.window_size[1] - 1))

    def _desc(self) -> L.AttnDesc:
        return L.AttnDesc(self.in_out_dims, self.num_heads, self.dims_per_head, self.window_size[0],
                          self.window_size[1], int(bool(self.use_cyclic_shift)))

    def _params(self) -> L.AttnParams:
        return L.AttnParams(_lin(self.q_for_heads), _lin(self.k_for_heads), _lin(self.v_for_heads),
                            _lin(self.linear_projection), _ptr(self.relative_position_bias_table))

    def _check_dropout(self):
        if self.training and (self.attention_drop_ratio or self.linear_after_att_drop_ratio):
            raise NotImplementedError("dropout > 0 in training mode is outside the forward-only HIP path")

    def forward(self, q: Tensor, k: Tensor, v: Tensor) -> Tensor:
        """a001:448-474.  With torch.autograd recording the call is differentiable (swf_window_attention_bwd: exact fp32)."""
        if _wants_grad(self, q, k, v):
            self._check_dropout()
            names = ("q_for_heads", "k_for_heads", "v_for_heads", "linear_projection")
            prm = [t for n in names for t in (getattr(self, n).weight, getattr(self, n).bias)]
            return _WindowAttentionFunction.apply(self, q, k, v, self.relative_position_bias_table, *prm)
        return self._forward_nograd(q, k, v)

    def _forward_nograd(self, q: Tensor, k: Tensor, v: Tensor) -> Tensor:
        _check_forward_only(self, q, k, v)
        self._check_dropout()
        if q.shape != k.shape or q.shape != v.shape:
            raise ValueError(f"q, k, v must share one shape, got {tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
        b, c, h, w = q.shape
        if c != self.in_out_dims:
            raise RuntimeError(f"expected {self.in_out_dims} channels, got {c}")
        self.feature_shape_hw = (h, w)
        qn = _to_nhwc(q)
        kn = qn if k is q else _to_nhwc(k)
        vn = kn if v is k else (qn if v is q else _to_nhwc(v))
        out = torch.empty((b, h, w, c), dtype=torch.float32, device=q.device)
        desc = self._desc()
        lib = L.lib()
        ws, wsn = _workspace(lib.swf_window_attention_workspace_bytes(C.byref(desc), b, h, w), q.device)
        prm = self._params()
        L.check(lib.swf_window_attention_fwd_prec(C.byref(desc), _precision_code(self.precision), C.byref(prm), _ptr(qn), _ptr(kn),
                                                  _ptr(vn), None, _ptr(out), b, h, w, ws, wsn, _stream(q.device)))
        return _to_nchw(out)


class _WindowAttentionFunction(torch.autograd.Function):
    """WindowAttention.forward under torch.autograd: forward = the module's own forward (its precision tier), backward =
    swf_window_attention_bwd.  q, k, v may be one tensor: autograd adds the three input gradients."""

    @staticmethod
    def forward(ctx, module, q, k, v, table, *params):
        ctx.module = module
        ctx.save_for_backward(q, k, v)
        with torch.no_grad():
            return module._forward_nograd(q.detach(), k.detach(), v.detach())

    @staticmethod
    def backward(ctx, g):
        m = ctx.module
        q, k, v = ctx.saved_tensors
        b, c, h, w = q.shape
        dev = q.device
        with torch.no_grad():
            qn = _to_nhwc(q)
            kn = qn if k is q else _to_nhwc(k)
            vn = kn if v is k else (qn if v is q else _to_nhwc(v))
            gn = _to_nhwc(g.contiguous())
            gq, gk, gv = torch.empty_like(qn), torch.empty_like(qn), torch.empty_like(qn)
            new = lambda t: None if t is None else torch.empty(t.shape, dtype=torch.float32, device=dev)
            lin = lambda wt, bs: L.Linear(wt.data_ptr(), None if bs is None else bs.data_ptr())
            mods = (m.q_for_heads, m.k_for_heads, m.v_for_heads, m.linear_projection)
            gw = [new(md.weight) for md in mods]
            gb = [new(md.bias) for md in mods]
            gt = new(m.relative_position_bias_table)
            grads = L.AttnParams(lin(gw[0], gb[0]), lin(gw[1], gb[1]), lin(gw[2], gb[2]), lin(gw[3], gb[3]), gt.data_ptr())
            lib, desc, prm = L.lib(), m._desc(), m._params()
            ws, wsn = _workspace(lib.swf_window_attention_bwd_workspace_bytes(C.byref(desc), b, h, w), dev)
            L.check(lib.swf_window_attention_bwd(C.byref(desc), C.byref(prm), _ptr(qn), _ptr(kn), _ptr(vn), _ptr(gn), _ptr(gq), _ptr(gk), _ptr(gv),
                                                 C.byref(grads), b, h, w, ws, wsn, _stream(dev)))
            flat = [t for pair in zip(gw, gb) for t in pair]
            return (None, _to_nchw(gq), _to_nchw(gk), _to_nchw(gv), gt, *flat)


class _MlpFunction(torch.autograd.Function):
    """One stream of AutoPathMLP.forward (a003:46-50) under torch.autograd: forward swf_mlp_fwd, backward swf_mlp_bwd."""

    @staticmethod
    def forward(ctx, module, s, x, w1, b1, w2, b2):
        ctx.module, ctx.s = module, s
        ctx.save_for_backward(x)
        with torch.no_grad():
            return module._one_nograd(x.detach(), s)

    @staticmethod
    def backward(ctx, g):
        m, s = ctx.module, ctx.s
        (x,) = ctx.saved_tensors
        b, c, h, w = x.shape
        dev = x.device
        c1, c2 = getattr(m, f"mlp_{s}_1"), getattr(m, f"mlp_{s}_2")
        with torch.no_grad():
            xn, gn = _to_nhwc(x), _to_nhwc(g.contiguous())
            gx = torch.empty_like(xn)
            new = lambda t: None if t is None else torch.empty(t.shape, dtype=torch.float32, device=dev)
            g1w, g1b, g2w, g2b = new(c1.weight), new(c1.bias), new(c2.weight), new(c2.bias)
            ptr = lambda t: None if t is None else t.data_ptr()
            f1, f2 = _lin(c1), _lin(c2)
            gf1, gf2 = L.Linear(ptr(g1w), ptr(g1b)), L.Linear(ptr(g2w), ptr(g2b))
            lib, n = L.lib(), b * h * w
            ws, wsn = _workspace(lib.swf_mlp_bwd_workspace_bytes(n, c, m.hidden_dims), dev)
            L.check(lib.swf_mlp_bwd(C.byref(f1), C.byref(f2), _ptr(xn), _ptr(gn), _ptr(gx), C.byref(gf1), C.byref(gf2), n, c, m.hidden_dims,
                                    ws, wsn, _stream(dev)))
            return None, None, _to_nchw(gx), g1w, g1b, g2w, g2b


class _LayerNormFunction(torch.autograd.Function):
    """my_layer_norm (a004:54-72: LayerNorm over the channels of an NCHW map) under torch.autograd."""

    @staticmethod
    def forward(ctx, ln, x, gamma, beta):
        ctx.ln = ln
        ctx.save_for_backward(x)
        b, c, h, w = x.shape
        xn = _to_nhwc(x.detach())
        out = torch.empty_like(xn)
        nrm = _norm(ln)
        L.check(L.lib().swf_layernorm_fwd(C.byref(nrm), _ptr(xn), _ptr(out), b * h * w, c, 0, _stream(x.device)))
        return _to_nchw(out)

    @staticmethod
    def backward(ctx, g):
        ln = ctx.ln
        (x,) = ctx.saved_tensors
        b, c, h, w = x.shape
        dev = x.device
        with torch.no_grad():
            xn, gn = _to_nhwc(x), _to_nhwc(g.contiguous())
            gx = torch.empty_like(xn)
            gg, gb = torch.empty_like(ln.weight), torch.empty_like(ln.bias)
            nrm, grads = _norm(ln), L.Norm(gg.data_ptr(), gb.data_ptr())
            lib, n = L.lib(), b * h * w
            ws, wsn = _workspace(lib.swf_layernorm_bwd_workspace_bytes(n, c), dev)
            L.check(lib.swf_layernorm_bwd(C.byref(nrm), _ptr(xn), _ptr(gn), _ptr(gx), C.byref(grads), n, c, ws, wsn, _stream(dev)))
            return None, _to_nchw(gx), gg, gb


# ----------------------------------------------------------------------------------------------
# a002 / a003: dual-stream routing of attention and MLP
# ----------------------------------------------------------------------------------------------
class AutoPathWinAtt(_FwdAlias, nn.Module):
    """a002_AutoPathWinAtt.AutoPathWinAtt: two WindowAttentions, self or cross routing (a002:58-82)."""

    def __init__(self, in_out_dims: int, num_heads: int, dims_per_head: int, window_size: tuple,
                 use_cyclic_shift: bool, use_dual_path: bool, use_cross_att: bool, use_qkv_bias: bool,
                 attention_drop_ratio: float, linear_after_att_drop_ratio: float):
        super().__init__()
        self.in_out_dims, self.num_heads, self.dims_per_head = in_out_dims, num_heads, dims_per_head
        self.window_size = tuple(window_size)
        self.use_cyclic_shift, self.use_dual_path, self.use_cross_att = use_cyclic_shift, use_dual_path, use_cross_att
        self.use_qkv_bias = use_qkv_bias
        self.attention_drop_ratio, self.linear_after_att_drop_ratio = attention_drop_ratio, linear_after_att_drop_ratio
        mk = lambda: WindowAttention(in_out_dims, num_heads, dims_per_head, window_size, use_cyclic_shift,
                                     use_cross_att, use_qkv_bias, attention_drop_ratio, linear_after_att_drop_ratio)
        self.window_attention_x = mk()
        if use_dual_path:
            self.window_attention_y = mk()

    def forward(self, x, y):
        if not self.use_dual_path:
            return self.window_attention_x(q=x, k=x, v=x)
        if self.use_cross_att:
            return self.window_attention_x(q=x, k=y, v=y), self.window_attention_y(q=y, k=x, v=x)
        return self.window_attention_x(q=x, k=x, v=x), self.window_attention_y(q=y, k=y, v=y)


class AutoPathMLP(_FwdAlias, nn.Module):
    """a003_AutoPathMLP.AutoPathMLP: per-stream 1x1 conv -> activation -> 1x1 conv (a003:21-50)."""

    def __init__(self, in_out_dims: int, hidden_dims: int, activation_func: nn.Module, use_dual_path: bool,
                 drop_ratio: float):
        super().__init__()
        self.in_out_dims, self.hidden_dims, self.activation_func = in_out_dims, hidden_dims, activation_func
        self.use_dual_path, self.drop_ratio = use_dual_path, drop_ratio
        self.precision = "fast"
        for s in ("x", "y") if use_dual_path else ("x",):
            c1 = nn.Conv2d(in_out_dims, hidden_dims, kernel_size=1)
            c2 = nn.Conv2d(hidden_dims, in_out_dims, kernel_size=1)
            d1, d2 = nn.Dropout(p=drop_ratio), nn.Dropout(p=drop_ratio)
            setattr(self, f"mlp_{s}_1", c1)
            setattr(self, f"mlp_{s}_2", c2)
            setattr(self, f"dropout_{s}_1", d1)
            setattr(self, f"dropout_{s}_2", d2)
            setattr(self, f"sequence_{s}", nn.Sequential(c1, activation_func, d1, c2, d2))

    def _params(self, s: str) -> L.BlockStreamParams:
        p = L.BlockStreamParams()
        p.fc1, p.fc2 = _lin(getattr(self, f"mlp_{s}_1")), _lin(getattr(self, f"mlp_{s}_2"))
        return p

    def _one_nograd(self, x: Tensor, s: str) -> Tensor:
        b, c, h, w = x.shape
        xn = _to_nhwc(x)
        ox = torch.empty_like(xn)
        lib, prec, n = L.lib(), _precision_code(self.precision), b * h * w
        px = self._params(s)
        ws, wsn = _workspace(lib.swf_mlp_workspace_bytes(prec, n, c, self.hidden_dims), x.device)
        L.check(lib.swf_mlp_fwd(prec, C.byref(px), None, _ptr(xn), None, _ptr(ox), None, n, c, self.hidden_dims, ws, wsn, _stream(x.device)))
        return _to_nchw(ox)

    def _one_grad(self, x: Tensor, s: str) -> Tensor:
        c1, c2 = getattr(self, f"mlp_{s}_1"), getattr(self, f"mlp_{s}_2")
        return _MlpFunction.apply(self, s, x, c1.weight, c1.bias, c2.weight, c2.bias)

    def forward(self, x, y=None):
        """a003:46-50 through swf_mlp_fwd: both streams in one call (one launch of the fused kernel's MLP half at level-0 width).  With
        torch.autograd recording each stream is a differentiable call (swf_mlp_bwd: exact fp32)."""
        if _wants_grad(self, x, y):
            _require_elu(self.activation_func)
            if self.training and self.drop_ratio:
                raise NotImplementedError("dropout > 0 in training mode is outside the HIP path")
            if self.use_dual_path or y is not None:
                return self._one_grad(x, "x"), self._one_grad(y, "y")
            return self._one_grad(x, "x")
        _check_forward_only(self, x, y)
        _require_elu(self.activation_func)
        if self.training and self.drop_ratio:
            raise NotImplementedError("dropout > 0 in training mode is outside the forward-only HIP path")
        dual = self.use_dual_path or y is not None
        b, c, h, w = x.shape
        xn, yn = _to_nhwc(x), (_to_nhwc(y) if dual else None)
        ox = torch.empty_like(xn)
        oy = torch.empty_like(yn) if dual else None
        lib, prec, n = L.lib(), _precision_code(self.precision), b * h * w
        px, py = self._params("x"), (self._params("y") if dual else None)
        ws, wsn = _workspace(lib.swf_mlp_workspace_bytes(prec, n, c, self.hidden_dims), x.device)
        L.check(lib.swf_mlp_fwd(prec, C.byref(px), C.byref(py) if dual else None, _ptr(xn), _ptr(yn) if dual else None, _ptr(ox),
                                _ptr(oy) if dual else None, n, c, self.hidden_dims, ws, wsn, _stream(x.device)))
        if dual:
            return _to_nchw(ox), _to_nchw(oy)
        return _to_nchw(ox)


# ----------------------------------------------------------------------------------------------
# a004: pre-norm residual wrapper
# ----------------------------------------------------------------------------------------------
class AddAndLayerNormWithOtherModule(_FwdAlias, nn.Module):
    """a004: out = x + other(LN_x(x), LN_y(y)) per stream (a004:20-48).  `other_module` must be one of
    this package's AutoPathWinAtt / AutoPathMLP — those are the two uses in the reference (a005:70-82) and
    the two fused half-block units of the C-ABI."""

    def __init__(self, normalized_shape: list, use_dual_path: bool, other_module: nn.Module):
        super().__init__()
        self.normalized_shape, self.use_dual_path, self.other_module = normalized_shape, use_dual_path, other_module
        self.precision = "fast"
        self.norm_layer_1 = nn.LayerNorm(normalized_shape=normalized_shape)
        if use_dual_path:
            self.norm_layer_2 = nn.LayerNorm(normalized_shape=normalized_shape)

    def forward(self, x, y=None):
        om = self.other_module
        dual = self.use_dual_path or y is not None
        if _wants_grad(self, x, y):
            # under torch.autograd the wrapper is composed of differentiable calls: LayerNorm, the other module, the residual add
            if not isinstance(om, (AutoPathWinAtt, AutoPathMLP)):
                raise NotImplementedError("other_module must be AutoPathWinAtt or AutoPathMLP of this package")
            nx = _LayerNormFunction.apply(self.norm_layer_1, x, self.norm_layer_1.weight, self.norm_layer_1.bias)
            if not dual:
                return _AddFunction.apply(x, om(nx, None) if isinstance(om, AutoPathMLP) else om(nx, nx))
            ny = _LayerNormFunction.apply(self.norm_layer_2, y, self.norm_layer_2.weight, self.norm_layer_2.bias)
            ox, oy = om(nx, ny)
            return _AddFunction.apply(x, ox), _AddFunction.apply(y, oy)
        _check_forward_only(self, x, y)
        b, c, h, w = x.shape
        xn, yn = _to_nhwc(x), (_to_nhwc(y) if dual else None)
        ox = torch.empty_like(xn)
        oy = torch.empty_like(yn) if dual else None
        lib = L.lib()
        streams = ("x", "y") if dual else ("x",)
        prec = _precision_code(self.precision)
        norms = {"x": self.norm_layer_1, "y": getattr(self, "norm_layer_2", None)}
        prm = {}
        if isinstance(om, AutoPathWinAtt):
            wa0 = om.window_attention_x
            desc = L.BlockDesc(wa0._desc(), 1, int(bool(om.use_cross_att)), prec)
            for s in streams:
                p = L.BlockStreamParams()
                p.ln1 = _norm(norms[s])
                p.attn = getattr(om, f"window_attention_{s}")._params()
                prm[s] = p
            fn = lib.swf_attn_halfblock_fwd
        elif isinstance(om, AutoPathMLP):
            _require_elu(om.activation_func)
            desc = L.BlockDesc(L.AttnDesc(om.in_out_dims, 1, 1, 1, 1, 0), om.hidden_dims, 0, prec)
            for s in streams:
                p = L.BlockStreamParams()
                p.ln2 = _norm(norms[s])
                p.fc1, p.fc2 = _lin(getattr(om, f"mlp_{s}_1")), _lin(getattr(om, f"mlp_{s}_2"))
                prm[s] = p
            fn = lib.swf_mlp_halfblock_fwd
        else:
            raise NotImplementedError("other_module must be AutoPathWinAtt or AutoPathMLP of this package")
        ws, wsn = _workspace(lib.swf_basic_block_workspace_bytes(
            C.byref(L.BlockDesc(desc.attn, max(desc.hidden, 1), 0, prec)), b, h, w), x.device)
        L.check(fn(C.byref(desc), C.byref(prm["x"]), C.byref(prm["y"]) if dual else None, _ptr(xn),
                   _ptr(yn) if dual else None, _ptr(ox), _ptr(oy) if dual else None, b, h, w, ws, wsn, _stream(x.device)))
        if dual:
            return _to_nchw(ox), _to_nchw(oy)
        return _to_nchw(ox)


# ----------------------------------------------------------------------------------------------
# a005 / a009 / a012: blocks
# ----------------------------------------------------------------------------------------------
class _BasicBlockFunction(torch.autograd.Function):
    """BasicBlock under torch.autograd: forward = the library's forward, backward = swf_basic_block_bwd (kernels_bwd.hip)."""

    @staticmethod
    def forward(ctx, block, dual, x, y, *params):
        ctx.block, ctx.dual = block, dual
        ctx.save_for_backward(x, y) if dual else ctx.save_for_backward(x)
        ctx.nparams = len(params)
        with torch.no_grad():
            ox, oy = block._forward_nograd(x.detach(), y.detach() if dual else None, dual)
        return (ox, oy) if dual else (ox,)

    @staticmethod
    def backward(ctx, *gouts):
        block, dual = ctx.block, ctx.dual
        saved = ctx.saved_tensors
        x, y = saved[0], (saved[1] if dual else None)
        with torch.no_grad():
            gx, gy, bufs = block._backward(x, y, gouts[0], gouts[1] if dual else None, dual)
        streams = ("x", "y") if dual else ("x",)
        pg = [g for s in streams for g in bufs[s]]
        return (None, None, gx, gy, *pg)


class BasicBlock(_FwdAlias, nn.Module):
    """a005_BasicBlock.BasicBlock (ctor a005:11-28, forward a005:127-145)."""

    def __init__(self, in_out_dims: int, num_heads: int, dims_per_head: int, window_size: tuple,
                 use_cyclic_shift: bool, use_dual_path: bool, use_cross_attr: bool, use_qkv_bias: bool,
                 attention_drop_ratio: float, linear_after_att_drop_ratio: float, mlp_hidden_dims: int,
                 mlp_activation_func: nn.Module, mlp_drop_ratio: float):
        super().__init__()
        self.in_out_dims, self.num_heads, self.dims_per_head = in_out_dims, num_heads, dims_per_head
        self.window_size = tuple(window_size)
        self.use_cyclic_shift, self.use_dual_path, self.use_cross_attr = use_cyclic_shift, use_dual_path, use_cross_attr
        self.use_qkv_bias = use_qkv_bias
        self.attention_drop_ratio, self.linear_after_att_drop_ratio = attention_drop_ratio, linear_after_att_drop_ratio
        self.mlp_hidden_dims, self.mlp_activation_func, self.mlp_drop_ratio = mlp_hidden_dims, mlp_activation_func, mlp_drop_ratio
        self.input_compatibility_with_cross_option = None
        self.precision = "fast"
        self.auto_path_win_att = AutoPathWinAtt(in_out_dims, num_heads, dims_per_head, window_size, use_cyclic_shift,
                                                use_dual_path, use_cross_attr, use_qkv_bias, attention_drop_ratio,
                                                linear_after_att_drop_ratio)
        self.auto_path_mlp = AutoPathMLP(in_out_dims, mlp_hidden_dims, mlp_activation_func, use_dual_path, mlp_drop_ratio)
        # the same sub-modules registered a second time, as in the reference (a005:70-82) -> aliased state_dict keys
        self.stage_1 = AddAndLayerNormWithOtherModule([in_out_dims], use_dual_path, self.auto_path_win_att)
        self.stage_2 = AddAndLayerNormWithOtherModule([in_out_dims], use_dual_path, self.auto_path_mlp)

    def check_input_compatibility_with_option(self, x, y):
        """First-call sanity check of a005:88-125.  The reference prints and calls exit(); this
        raises ValueError instead (documented deviation)."""
        if self.input_compatibility_with_cross_option is not None:
            return
        ok = x is not None and (y is not None) == bool(self.use_dual_path)
        if ok and self.use_cross_attr and y is not None and torch.equal(x, y):
            ok = False
        if not ok:   # stays None: every later call re-checks and raises again (the reference would have exited, a005:118)
            raise ValueError("inputs are incompatible with the cross_attr / dual_path options "
                             "(y missing or unexpected, or cross attention given identical x and y)")
        self.input_compatibility_with_cross_option = True

    def _desc(self, precision) -> L.BlockDesc:
        return L.BlockDesc(self.auto_path_win_att.window_attention_x._desc(), self.mlp_hidden_dims,
                           int(bool(self.use_cross_attr)), _precision_code(precision))

    def _stream_params(self, s: str) -> L.BlockStreamParams:
        _require_elu(self.mlp_activation_func)
        p = L.BlockStreamParams()
        idx = "1" if s == "x" else "2"
        p.ln1 = _norm(getattr(self.stage_1, f"norm_layer_{idx}"))
        p.attn = getattr(self.auto_path_win_att, f"window_attention_{s}")._params()
        p.ln2 = _norm(getattr(self.stage_2, f"norm_layer_{idx}"))
        p.fc1, p.fc2 = _lin(getattr(self.auto_path_mlp, f"mlp_{s}_1")), _lin(getattr(self.auto_path_mlp, f"mlp_{s}_2"))
        return p

    # ---- training side (first stage): the block under torch.autograd --------------------------------------------
    _GRAD_FIELDS = ("ln1.weight", "ln1.bias", "q.weight", "q.bias", "k.weight", "k.bias", "v.weight", "v.bias", "proj.weight",
                    "proj.bias", "table", "ln2.weight", "ln2.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")

    def _grad_tensors(self, s: str) -> List[Optional[Tensor]]:
        """The stream's parameter tensors in the order of _GRAD_FIELDS (None where a layer has no bias)."""
        idx = "1" if s == "x" else "2"
        wa = getattr(self.auto_path_win_att, f"window_attention_{s}")
        ln1, ln2 = getattr(self.stage_1, f"norm_layer_{idx}"), getattr(self.stage_2, f"norm_layer_{idx}")
        f1, f2 = getattr(self.auto_path_mlp, f"mlp_{s}_1"), getattr(self.auto_path_mlp, f"mlp_{s}_2")
        return [ln1.weight, ln1.bias, wa.q_for_heads.weight, wa.q_for_heads.bias, wa.k_for_heads.weight, wa.k_for_heads.bias,
                wa.v_for_heads.weight, wa.v_for_heads.bias, wa.linear_projection.weight, wa.linear_projection.bias,
                wa.relative_position_bias_table, ln2.weight, ln2.bias, f1.weight, f1.bias, f2.weight, f2.bias]

    @staticmethod
    def _grads_struct(bufs: List[Optional[Tensor]]) -> L.BlockStreamParams:
        """swf_block_stream_grads over freshly allocated gradient buffers (same field order as swf_block_stream_params)."""
        g = L.BlockStreamParams()
        ptr = lambda t: None if t is None else t.data_ptr()
        g.ln1 = L.Norm(ptr(bufs[0]), ptr(bufs[1]))
        g.attn = L.AttnParams(L.Linear(ptr(bufs[2]), ptr(bufs[3])), L.Linear(ptr(bufs[4]), ptr(bufs[5])), L.Linear(ptr(bufs[6]), ptr(bufs[7])),
                              L.Linear(ptr(bufs[8]), ptr(bufs[9])), ptr(bufs[10]))
        g.ln2 = L.Norm(ptr(bufs[11]), ptr(bufs[12]))
        g.fc1, g.fc2 = L.Linear(ptr(bufs[13]), ptr(bufs[14])), L.Linear(ptr(bufs[15]), ptr(bufs[16]))
        return g

    def _forward_nograd(self, x, y, dual):
        b, c, h, w = x.shape
        xn, yn = _to_nhwc(x), (_to_nhwc(y) if dual else None)
        ox = torch.empty_like(xn)
        oy = torch.empty_like(yn) if dual else None
        desc = self._desc(self.precision)
        px = self._stream_params("x")
        py = self._stream_params("y") if dual else None
        lib = L.lib()
        ws, wsn = _workspace(lib.swf_basic_block_workspace_bytes(C.byref(desc), b, h, w), x.device)
        L.check(lib.swf_basic_block_fwd(C.byref(desc), C.byref(px), C.byref(py) if dual else None, _ptr(xn),
                                        _ptr(yn) if dual else None, _ptr(ox), _ptr(oy) if dual else None,
                                        b, h, w, ws, wsn, _stream(x.device)))
        return (_to_nchw(ox), _to_nchw(oy)) if dual else (_to_nchw(ox), None)

    def _backward(self, x, y, gox, goy, dual):
        """dL/d(x, y) and the parameter gradients of both streams through swf_basic_block_bwd (exact fp32, forward recomputed)."""
        b, c, h, w = x.shape
        dev = x.device
        xn, yn = _to_nhwc(x), (_to_nhwc(y) if dual else None)
        zeros = lambda t: torch.zeros_like(t) if t is None else t
        gxo = _to_nhwc(gox.contiguous() if gox is not None else torch.zeros_like(x))
        gyo = _to_nhwc(goy.contiguous() if goy is not None else torch.zeros_like(y)) if dual else None
        gxi, gyi = torch.empty_like(xn), (torch.empty_like(yn) if dual else None)
        streams = ("x", "y") if dual else ("x",)
        bufs = {s: [None if t is None else torch.empty(t.shape, dtype=torch.float32, device=dev) for t in self._grad_tensors(s)] for s in streams}
        gs = {s: self._grads_struct(bufs[s]) for s in streams}
        desc = self._desc("fp32")
        px = self._stream_params("x")
        py = self._stream_params("y") if dual else None
        lib = L.lib()
        ws, wsn = _workspace(lib.swf_basic_block_bwd_workspace_bytes(C.byref(desc), b, h, w), dev)
        L.check(lib.swf_basic_block_bwd(C.byref(desc), C.byref(px), C.byref(py) if dual else None, _ptr(xn), _ptr(yn) if dual else None,
                                        _ptr(gxo), _ptr(gyo) if dual else None, _ptr(gxi), _ptr(gyi) if dual else None,
                                        C.byref(gs["x"]), C.byref(gs["y"]) if dual else None, b, h, w, ws, wsn, _stream(dev)))
        return _to_nchw(gxi), (_to_nchw(gyi) if dual else None), bufs

    def forward(self, x, y=None):
        dual = self.use_dual_path or y is not None
        if torch.is_grad_enabled() and (x.requires_grad or (y is not None and y.requires_grad) or
                                        any(p.requires_grad for p in self.parameters())):
            # autograd path: forward through the library as usual, backward through swf_basic_block_bwd (first stage of the training
            # side: the block; patch layers, head and loss have no backward yet, so MyModel as a whole still raises)
            for t in (x, y):
                if t is not None and t.dim() != 4:
                    raise ValueError(f"expected a 4-D (batch, channels, height, width) tensor, got shape {tuple(t.shape)}")
            if self.training and (self.attention_drop_ratio or self.linear_after_att_drop_ratio or self.mlp_drop_ratio):
                raise NotImplementedError("dropout > 0 in training mode is outside the HIP path")
            _require_elu(self.mlp_activation_func)
            self.check_input_compatibility_with_option(x=x, y=y)
            streams = ("x", "y") if dual else ("x",)
            params = [t for s in streams for t in self._grad_tensors(s)]
            out = _BasicBlockFunction.apply(self, dual, x, y if dual else None, *params)
            return (out[0], out[1]) if dual else out[0]
        _check_forward_only(self, x, y)
        self.check_input_compatibility_with_option(x=x, y=y)
        b, c, h, w = x.shape
        xn, yn = _to_nhwc(x), (_to_nhwc(y) if dual else None)
        ox = torch.empty_like(xn)
        oy = torch.empty_like(yn) if dual else None
        desc = self._desc(self.precision)
        px = self._stream_params("x")
        py = self._stream_params("y") if dual else None
        lib = L.lib()
        ws, wsn = _workspace(lib.swf_basic_block_workspace_bytes(C.byref(desc), b, h, w), x.device)
        L.check(lib.swf_basic_block_fwd(C.byref(desc), C.byref(px), C.byref(py) if dual else None, _ptr(xn),
                                        _ptr(yn) if dual else None, _ptr(ox), _ptr(oy) if dual else None,
                                        b, h, w, ws, wsn, _stream(x.device)))
        return (_to_nchw(ox), _to_nchw(oy)) if dual else _to_nchw(ox)


class NormalAndShiftWinsBlockPair(_FwdAlias, nn.Module):
    """a009: BasicBlock(shift=False) then BasicBlock(shift=True) (a009:57-109)."""

    def __init__(self, in_out_dims: int, num_heads: int, dims_per_head: int, window_size: tuple, use_dual_path: bool,
                 use_cross_attr: bool, use_qkv_bias: bool, attention_drop_ratio: float,
                 linear_after_att_drop_ratio: float, mlp_hidden_dims: int, mlp_activation_func: nn.Module,
                 mlp_drop_ratio: float):
        super().__init__()
        self.in_out_dims, self.num_heads, self.dims_per_head = in_out_dims, num_heads, dims_per_head
        self.window_size, self.use_dual_path, self.use_cross_attr = tuple(window_size), use_dual_path, use_cross_attr
        self.use_qkv_bias = use_qkv_bias
        self.attention_drop_ratio, self.linear_after_att_drop_ratio = attention_drop_ratio, linear_after_att_drop_ratio
        self.mlp_hidden_dims, self.mlp_activation_func, self.mlp_drop_ratio = mlp_hidden_dims, mlp_activation_func, mlp_drop_ratio
        mk = lambda shift: BasicBlock(in_out_dims, num_heads, dims_per_head, window_size, shift, use_dual_path,
                                      use_cross_attr, use_qkv_bias, attention_drop_ratio, linear_after_att_drop_ratio,
                                      mlp_hidden_dims, mlp_activation_func, mlp_drop_ratio)
        self.normal_window_block = mk(False)
        self.shifted_window_block = mk(True)

    def forward(self, x, y=None):
        if self.use_dual_path:
            x, y = self.normal_window_block(x=x, y=y)
            return self.shifted_window_block(x=x, y=y)
        x = self.normal_window_block(x=x, y=None)
        return self.shifted_window_block(x=x, y=None)


class SelfAndCrossBlockPair(_FwdAlias, nn.Module):
    """Drop-in for a012_SelfAndCrossBlockPair.SelfAndCrossBlockPair (ctor a012:10-25, forward a012:70-78):
    four BasicBlocks — self/normal, self/shifted, cross/normal, cross/shifted — as ONE C-ABI call."""

    def __init__(self, in_out_dims: int, num_heads: int, dims_per_head: int, window_size: tuple, use_dual_path: bool,
                 use_qkv_bias: bool, attention_drop_ratio: float, linear_after_att_drop_ratio: float,
                 mlp_hidden_dims: int, mlp_activation_func: nn.Module, mlp_drop_ratio: float):
        super().__init__()
        self.in_out_dims, self.num_heads, self.dims_per_head = in_out_dims, num_heads, dims_per_head
        self.window_size, self.use_dual_path, self.use_qkv_bias = tuple(window_size), use_dual_path, use_qkv_bias
        self.attention_drop_ratio, self.linear_after_att_drop_ratio = attention_drop_ratio, linear_after_att_drop_ratio
        self.mlp_hidden_dims, self.mlp_activation_func, self.mlp_drop_ratio = mlp_hidden_dims, mlp_activation_func, mlp_drop_ratio
        self.precision = "fast"
        mk = lambda cross: NormalAndShiftWinsBlockPair(in_out_dims, num_heads, dims_per_head, window_size, use_dual_path,
                                                       cross, use_qkv_bias, attention_drop_ratio,
                                                       linear_after_att_drop_ratio, mlp_hidden_dims, mlp_activation_func,
                                                       mlp_drop_ratio)
        self.self_att_block = mk(False)
        self.cross_att_block = mk(True)

    def _blocks(self) -> List[BasicBlock]:
        return [self.self_att_block.normal_window_block, self.self_att_block.shifted_window_block,
                self.cross_att_block.normal_window_block, self.cross_att_block.shifted_window_block]

    def forward(self, x, y=None):
        blocks = self._blocks()
        dual = self.use_dual_path
        if dual and y is None:
            raise ValueError("use_dual_path=True needs both x and y")
        if _wants_grad(self, x, y):   # under autograd the four blocks run one by one (each differentiable: BasicBlock.forward)
            for blk in blocks:
                if dual:
                    x, y = blk(x, y)
                else:
                    x = blk(x)
            return (x, y) if dual else x
        _check_forward_only(self, x, y)
        blocks[2].check_input_compatibility_with_option(x=x, y=y if dual else None)
        b, c, h, w = x.shape
        xn, yn = _to_nhwc(x), (_to_nhwc(y) if dual else None)
        ox = torch.empty_like(xn)
        oy = torch.empty_like(yn) if dual else None
        desc = blocks[0]._desc(self.precision)
        px = (L.BlockStreamParams * 4)(*[blk._stream_params("x") for blk in blocks])
        py = (L.BlockStreamParams * 4)(*[blk._stream_params("y") for blk in blocks]) if dual else None
        lib = L.lib()
        ws, wsn = _workspace(lib.swf_basic_block_workspace_bytes(C.byref(desc), b, h, w), x.device)
        L.check(lib.swf_block_pair4_fwd(C.byref(desc), px, py, _ptr(xn), _ptr(yn) if dual else None, _ptr(ox),
                                        _ptr(oy) if dual else None, b, h, w, ws, wsn, _stream(x.device)))
        return (_to_nchw(ox), _to_nchw(oy)) if dual else _to_nchw(ox)


# ----------------------------------------------------------------------------------------------
# a006 / a011: padding and patch (un)merging
# ----------------------------------------------------------------------------------------------
class _ReflectPadFunction(torch.autograd.Function):
    """MyPadding encoder side under autograd: forward swf_reflect_pad_fwd, backward its adjoint swf_reflect_pad_bwd."""

    @staticmethod
    def forward(ctx, t, ph, pw):
        b, c, h, w = t.shape
        ctx.shape, ctx.pad = (b, c, h, w), (ph, pw)
        t = t.contiguous()
        out = torch.empty((b, c, h + ph, w + pw), dtype=torch.float32, device=t.device)
        L.check(L.lib().swf_reflect_pad_fwd(_ptr(t), _ptr(out), b * c, h, w, 1, ph, pw, _stream(t.device)))
        return out

    @staticmethod
    def backward(ctx, g):
        (b, c, h, w), (ph, pw) = ctx.shape, ctx.pad
        g = g.contiguous()
        dx = torch.empty((b, c, h, w), dtype=torch.float32, device=g.device)
        L.check(L.lib().swf_reflect_pad_bwd(_ptr(g), _ptr(dx), b * c, h, w, 1, ph, pw, _stream(g.device)))
        return dx, None, None


class _CropFunction(torch.autograd.Function):
    """MyPadding decoder side under autograd: forward swf_crop_fwd, backward the zero padding of the cropped rows / columns (a copy)."""

    @staticmethod
    def forward(ctx, t, ph, pw):
        b, c, h, w = t.shape
        ctx.shape = (b, c, h, w)
        t = t.contiguous()
        out = torch.empty((b, c, h - ph, w - pw), dtype=torch.float32, device=t.device)
        L.check(L.lib().swf_crop_fwd(_ptr(t), _ptr(out), b * c, h, w, h - ph, w - pw, 1, _stream(t.device)))
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, h, w = ctx.shape
        dx = torch.zeros((b, c, h, w), dtype=torch.float32, device=g.device)
        dx[:, :, : g.shape[2], : g.shape[3]].copy_(g)
        return dx, None, None


class _AddFunction(torch.autograd.Function):
    """The decoder's skip connection (a013:222-225) as a library call; both inputs receive the output gradient."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        L.check(L.lib().swf_add_fwd(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream(a.device)))
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class MyPadding(_FwdAlias, nn.Module):
    """a006_PaddingOperation.MyPadding: encoder side reflect-pads bottom/right to a multiple of
    `window_size` and pushes (shape, pad) on the shared recorders; decoder side pops and crops."""

    def __init__(self, belongs_to_encoder: bool, window_size: tuple, use_dual_path: bool,
                 feature_shape_recorder: StateRecorder, padding_size_recorder: StateRecorder):
        super().__init__()
        self.belongs_to_encoder, self.window_size, self.use_dual_path = belongs_to_encoder, tuple(window_size), use_dual_path
        self.feature_shape_hw: tuple = tuple()
        self.padding_size: tuple = tuple()
        self.feature_shape_recorder, self.padding_size_recorder = feature_shape_recorder, padding_size_recorder

    @staticmethod
    def calculate_padding_size(current_length, window_size):
        return (window_size - current_length % window_size) % window_size

    def _pad(self, t: Tensor) -> Tensor:
        ph, pw = self.padding_size
        if ph == 0 and pw == 0:
            return t
        if torch.is_grad_enabled() and t.requires_grad:
            if ph >= t.shape[2] or pw >= t.shape[3]:
                raise RuntimeError(f"reflect pad ({ph},{pw}) must be smaller than the map {tuple(t.shape[2:])} (a006:128)")
            return _ReflectPadFunction.apply(t, ph, pw)
        b, c, h, w = t.shape
        t = t.contiguous()
        out = torch.empty((b, c, h + ph, w + pw), dtype=torch.float32, device=t.device)
        # an NCHW tensor is B*C single-channel maps in the library's NHWC convention
        L.check(L.lib().swf_reflect_pad_fwd(_ptr(t), _ptr(out), b * c, h, w, 1, ph, pw, _stream(t.device)))
        return out

    def _crop(self, t: Tensor) -> Tensor:
        ph, pw = self.padding_size
        if ph == 0 and pw == 0:
            return t
        if torch.is_grad_enabled() and t.requires_grad:
            return _CropFunction.apply(t, ph, pw)
        b, c, h, w = t.shape
        t = t.contiguous()
        out = torch.empty((b, c, h - ph, w - pw), dtype=torch.float32, device=t.device)
        L.check(L.lib().swf_crop_fwd(_ptr(t), _ptr(out), b * c, h, w, h - ph, w - pw, 1, _stream(t.device)))
        return out

    def forward(self, x, y):
        if not _wants_grad(self, x, y):
            _check_forward_only(self, x, y)
        if self.belongs_to_encoder:
            h, w = x.shape[-2:]
            if not (self.training and self.feature_shape_hw):   # a006:38-52: refreshed every call in eval
                self.feature_shape_hw = (h, w)
            self.feature_shape_recorder.record(self.feature_shape_hw)
            if not (self.training and self.padding_size):
                fh, fw = self.feature_shape_hw
                self.padding_size = (self.calculate_padding_size(fh, self.window_size[0]),
                                     self.calculate_padding_size(fw, self.window_size[1]))
            self.padding_size_recorder.record(self.padding_size)
            op = self._pad
        else:
            self.feature_shape_hw = self.feature_shape_recorder.read()
            self.padding_size = self.padding_size_recorder.read()
            op = self._crop
        if self.use_dual_path:
            return op(x), (op(y) if y is not None else None)
        return op(x)


class _PatchLayerFunction(torch.autograd.Function):
    """One stream of PatchMergingAndLinearLayer under autograd: forward through the library, backward = swf_patch_layer_bwd."""

    @staticmethod
    def forward(ctx, layer, s, t, cw, cb, lg, lb):
        ctx.layer, ctx.s = layer, s
        ctx.save_for_backward(t)
        with torch.no_grad():
            return layer._one(t.detach(), s)

    @staticmethod
    def backward(ctx, g):
        layer, s = ctx.layer, ctx.s
        (t,) = ctx.saved_tensors
        b, c, h, w = t.shape
        mh, mw = layer.merging_or_unmerging_size
        conv, ln = getattr(layer, f"mlp_layer_{s}"), getattr(layer, f"layer_norm_{s}")
        dev = t.device
        with torch.no_grad():
            tn, gn = _to_nhwc(t), _to_nhwc(g.contiguous())
            gin = torch.empty_like(tn)
            gw = torch.empty(conv.weight.shape, dtype=torch.float32, device=dev)
            gb = torch.empty(conv.bias.shape, dtype=torch.float32, device=dev) if conv.bias is not None else None
            gg, gbe = torch.empty_like(ln.weight), torch.empty_like(ln.bias)
            prm = L.PatchParams(_lin(conv), _norm(ln))
            grads = L.PatchParams(L.Linear(gw.data_ptr(), gb.data_ptr() if gb is not None else None), L.Norm(gg.data_ptr(), gbe.data_ptr()))
            lib, enc = L.lib(), int(layer.belongs_to_encoder)
            ws, wsn = _workspace(lib.swf_patch_layer_bwd_workspace_bytes(b, h, w, layer.in_dims, layer.out_dims, mh, mw, enc), dev)
            L.check(lib.swf_patch_layer_bwd(C.byref(prm), _ptr(tn), _ptr(gn), _ptr(gin), C.byref(grads), b, h, w, layer.in_dims, layer.out_dims,
                                            mh, mw, enc, ws, wsn, _stream(dev)))
            return None, None, _to_nchw(gin), gw, gb, gg, gbe


class PatchMergingAndLinearLayer(_FwdAlias, nn.Module):
    """a011_PatchOperation.PatchMergingAndLinearLayer (ctor a011:26-70, forward a011:244-264)."""

    def __init__(self, belongs_to_encoder: bool, use_dual_path: bool, in_dims: int, out_dims: int,
                 patch_merging_size_recorder: StateRecorder, merging_or_unmerging_size: tuple,
                 activation_func: nn.Module = nn.ELU()):
        super().__init__()
        self.belongs_to_encoder, self.use_dual_path = belongs_to_encoder, use_dual_path
        self.in_dims, self.out_dims = in_dims, out_dims
        self.patch_merging_recorder = patch_merging_size_recorder
        self.merging_or_unmerging_size = tuple(merging_or_unmerging_size)
        self.activation_func = activation_func
        ratio = self.merging_or_unmerging_size[0] * self.merging_or_unmerging_size[1]
        self.conv_in_dims = in_dims * ratio if belongs_to_encoder else in_dims
        self.conv_out_dims = out_dims if belongs_to_encoder else out_dims * ratio
        self.mlp_layer_x = nn.Conv2d(self.conv_in_dims, self.conv_out_dims, kernel_size=1)
        self.layer_norm_x = nn.LayerNorm(normalized_shape=self.conv_out_dims)
        if use_dual_path:
            self.mlp_layer_y = nn.Conv2d(self.conv_in_dims, self.conv_out_dims, kernel_size=1)
            self.layer_norm_y = nn.LayerNorm(normalized_shape=self.conv_out_dims)
        self.register_buffer(name="buffer_to_show_device", tensor=torch.zeros(size=(1,)))

    def _one(self, t: Tensor, s: str) -> Tensor:
        _require_elu(self.activation_func)
        b, c, h, w = t.shape
        mh, mw = self.merging_or_unmerging_size
        prm = L.PatchParams(_lin(getattr(self, f"mlp_layer_{s}")), _norm(getattr(self, f"layer_norm_{s}")))
        tn = _to_nhwc(t)
        lib = L.lib()
        if self.belongs_to_encoder:
            if h % mh or w % mw:   # einops raises in the reference (a011:87-93); MyPadding runs first in the model
                raise ValueError(f"map {h}x{w} is not divisible by the merging size {mh}x{mw}")
            out = torch.empty((b, h // mh, w // mw, self.out_dims), dtype=torch.float32, device=t.device)
            ws, wsn = _workspace(lib.swf_patch_workspace_bytes(b, h, w, self.in_dims, self.out_dims, mh, mw, 1, 1, 1), t.device)
            L.check(lib.swf_patch_merge_fwd(C.byref(prm), _ptr(tn), _ptr(out), b, h, w, self.in_dims, self.out_dims,
                                            mh, mw, 1, 1, ws, wsn, _stream(t.device)))
        else:
            out = torch.empty((b, h * mh, w * mw, self.out_dims), dtype=torch.float32, device=t.device)
            ws, wsn = _workspace(lib.swf_patch_workspace_bytes(b, h, w, self.in_dims, self.out_dims, mh, mw, 1, 1, 0), t.device)
            L.check(lib.swf_patch_unmerge_fwd(C.byref(prm), _ptr(tn), None, _ptr(out), b, h, w, h, w, self.in_dims,
                                              self.out_dims, mh, mw, h * mh, w * mw, ws, wsn, _stream(t.device)))
        return _to_nchw(out)

    def _one_grad(self, t: Tensor, s: str) -> Tensor:
        conv, ln = getattr(self, f"mlp_layer_{s}"), getattr(self, f"layer_norm_{s}")
        return _PatchLayerFunction.apply(self, s, t, conv.weight, conv.bias, ln.weight, ln.bias)

    def forward(self, x, y=None):
        grad = _wants_grad(self, x, y)
        if not grad:
            _check_forward_only(self, x, y)
        if x.shape[1] != self.in_dims:
            raise RuntimeError(f"expected {self.in_dims} channels, got {x.shape[1]}")
        one = self._one_grad if grad else self._one
        if y is not None:
            return one(x, "x"), one(y, "y")
        return one(x, "x")


# ----------------------------------------------------------------------------------------------
# a013: model assembly
# ----------------------------------------------------------------------------------------------
class _HeadFunction(torch.autograd.Function):
    """MyModel.do_final_layer (a013:126-152) under autograd: forward swf_final_head_fwd (after swf_final_head_batch_stats in training
    mode), backward swf_final_head_bwd."""

    @staticmethod
    def forward(ctx, model, x, y, c1w, c1b, g, bt, c2w, c2b):
        ctx.model, ctx.train = model, bool(model.training)
        b, _, h, w = x.shape
        x, y = x.detach().contiguous(), y.detach().contiguous()
        out = torch.empty((b, 1, h, w), dtype=torch.float32, device=x.device)
        lib, hp, ks = L.lib(), model._head_params(), model.final_layer_conv_kernel_size
        ws, wsn = _workspace(lib.swf_final_head_bwd_workspace_bytes(b, h, w, ks), x.device)
        stats = torch.empty(4, dtype=torch.float32, device=x.device)   # batch mean[2], biased variance[2]
        if ctx.train:
            # nn.BatchNorm2d in training mode (the reference trains under model.train(), a016:137): normalise with the batch statistics
            # of this forward and move the running statistics towards them (momentum 0.1, unbiased variance)
            bn = model.final_layer[1]
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            L.check(lib.swf_final_head_batch_stats(C.byref(hp), _ptr(x), _ptr(y), stats.data_ptr(), stats.data_ptr() + 8,
                                                   _ptr(bn.running_mean) if bn.track_running_stats else None,
                                                   _ptr(bn.running_var) if bn.track_running_stats else None, mom, b, h, w, ks, ws, wsn, _stream(x.device)))
            if bn.track_running_stats and bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            hp.bn_mean, hp.bn_var = stats.data_ptr(), stats.data_ptr() + 8
        ctx.save_for_backward(x, y, stats)
        L.check(lib.swf_final_head_fwd(C.byref(hp), _ptr(x), _ptr(y), _ptr(out), b, h, w, ks, ws, wsn, _stream(x.device)))
        return out

    @staticmethod
    def backward(ctx, gout):
        model = ctx.model
        x, y, stats = ctx.saved_tensors
        b, _, h, w = x.shape
        dev = x.device
        conv1, bn, conv2 = model.final_layer[0], model.final_layer[1], model.final_layer[3]
        with torch.no_grad():
            x, y, gout = x.contiguous(), y.contiguous(), gout.contiguous()
            gx, gy = torch.empty_like(x), torch.empty_like(y)
            new = lambda t: None if t is None else torch.empty(t.shape, dtype=torch.float32, device=dev)
            g1w, g1b, gg, gb, g2w, g2b = new(conv1.weight), new(conv1.bias), new(bn.weight), new(bn.bias), new(conv2.weight), new(conv2.bias)
            ptr = lambda t: None if t is None else t.data_ptr()
            grads = L.HeadGrads(ptr(g1w), ptr(g1b), ptr(gg), ptr(gb), ptr(g2w), ptr(g2b))
            lib, hp, ks = L.lib(), model._head_params(), model.final_layer_conv_kernel_size
            if ctx.train:
                hp.bn_mean, hp.bn_var = stats.data_ptr(), stats.data_ptr() + 8
            ws, wsn = _workspace(lib.swf_final_head_bwd_workspace_bytes(b, h, w, ks), dev)
            L.check(lib.swf_final_head_bwd(C.byref(hp), _ptr(x), _ptr(y), _ptr(gout), _ptr(gx), _ptr(gy), C.byref(grads), b, h, w, ks,
                                           int(ctx.train), ws, wsn, _stream(dev)))
        return None, gx, gy, g1w, g1b, gg, gb, g2w, g2b


def get_encoder_or_decoder_block(mode: str, window_size: tuple, feature_shape_recorder: StateRecorder,
                                 padding_size_recorder: StateRecorder, merging_size: tuple, in_dims: int, out_dims: int,
                                 patch_merging_size_recorder: StateRecorder, att_num_heads: int, att_dims_per_head: int,
                                 attention_drop_ratio: float, linear_after_att_drop_ratio: float, mlp_hidden_dims: int,
                                 mlp_activation_func: nn.Module, mlp_drop_ratio: float) -> nn.ModuleList:
    """a013:236-314: [pad(merge), merge, pad(window), blocks] for the encoder, the same four in
    reverse for the decoder; attention width is out_dims (encoder) / in_dims (decoder)."""
    if mode not in ("encoder", "decoder"):
        raise ValueError("mode must be either encoder or decoder")
    enc = mode == "encoder"
    mods = [
        MyPadding(enc, merging_size, True, feature_shape_recorder, padding_size_recorder),
        PatchMergingAndLinearLayer(enc, True, in_dims, out_dims, patch_merging_size_recorder, merging_size,
                                   mlp_activation_func),
        MyPadding(enc, window_size, True, feature_shape_recorder, padding_size_recorder),
        SelfAndCrossBlockPair(out_dims if enc else in_dims, att_num_heads, att_dims_per_head, window_size, True, True,
                              attention_drop_ratio, linear_after_att_drop_ratio, mlp_hidden_dims, mlp_activation_func,
                              mlp_drop_ratio),
    ]
    return nn.ModuleList(mods if enc else mods[::-1])


class MyModel(_FwdAlias, nn.Module):
    """Drop-in for a013_ModelDefinition.MyModel (ctor a013:18-38, forward a013:209-230).

    `forward(in_x, in_y)` is ONE call into the C-ABI (`swf_model_forward`): the parameters are
    copied once into a flat device arena whose layout the library defines by state_dict key
    (`swf_model_param_info`), and the whole U-Net runs from that arena on the current stream.
    `precision` ('fast' | 'fp32') selects the arithmetic mode (include/swinfuse.h swf_precision); `schedule` ('latency' |
    'throughput', swf_schedule) the kernel shapes of the fast tier where it has a choice: 'latency' for one forward at a time,
    'throughput' for several forwards in flight (shard.ShardedFusion sets it for its lanes).
    """

    def __init__(self, window_size: tuple, merging_size: tuple, in_dims_list: list, out_dims_list: list,
                 att_num_heads: int, att_dims_per_head_ratio: float, attention_drop_ratio: float,
                 linear_after_att_drop_ratio: float, mlp_hidden_dims_ratio: int, mlp_activation_func: nn.Module,
                 mlp_drop_ratio: float, final_layer_att_dims_per_head_ratio: float, final_conv_layer_kernel_size: int,
                 final_layer_mlp_hidden_dims_ratio: int):
        super().__init__()
        self.window_size, self.merging_size = tuple(window_size), tuple(merging_size)
        self.in_dims_list, self.out_dims_list = list(in_dims_list), list(out_dims_list)
        self.att_num_heads, self.att_dims_per_head_ratio = att_num_heads, att_dims_per_head_ratio
        self.attention_drop_ratio, self.linear_after_att_drop_ratio = attention_drop_ratio, linear_after_att_drop_ratio
        self.mlp_hidden_dims_ratio, self.mlp_activation_func, self.mlp_drop_ratio = mlp_hidden_dims_ratio, mlp_activation_func, mlp_drop_ratio
        self.final_layer_att_dims_per_head_ratio = final_layer_att_dims_per_head_ratio
        self.final_layer_conv_kernel_size = final_conv_layer_kernel_size
        self.final_layer_mlp_hidden_dims_ratio = final_layer_mlp_hidden_dims_ratio
        self.feature_shape_recorder, self.padding_size_recorder = StateRecorder(), StateRecorder()
        self.patch_merging_size_recorder, self.u_net_intermediate_result_recorder = StateRecorder(), StateRecorder()
        self.precision = "fast"
        self.schedule = "latency"
        self._arena: Optional[Tensor] = None
        self._arena_key = None
        self._packed: Optional[Tensor] = None
        # bumped whenever the arena / packed images are dropped: anything that baked their addresses into a captured
        # hipGraph (shard.ShardedFusion) compares it before replaying
        self.weights_epoch = 0
        # the reference's first-forward input check (a005:98-118), None = not run yet
        self.input_compatibility_with_cross_option: Optional[bool] = None

        enc, dec = deque(), deque()
        for j in range(len(self.in_dims_list) - 1, -1, -1):   # a013:154-207
            common = dict(window_size=self.window_size, feature_shape_recorder=self.feature_shape_recorder,
                          padding_size_recorder=self.padding_size_recorder, merging_size=self.merging_size,
                          patch_merging_size_recorder=self.patch_merging_size_recorder, att_num_heads=att_num_heads,
                          att_dims_per_head=math.floor(self.out_dims_list[j] * att_dims_per_head_ratio),
                          attention_drop_ratio=attention_drop_ratio, linear_after_att_drop_ratio=linear_after_att_drop_ratio,
                          mlp_activation_func=mlp_activation_func, mlp_drop_ratio=mlp_drop_ratio)
            enc.appendleft(get_encoder_or_decoder_block(mode="encoder", in_dims=self.in_dims_list[j],
                                                        out_dims=self.out_dims_list[j],
                                                        mlp_hidden_dims=self.out_dims_list[j] * mlp_hidden_dims_ratio, **common))
            dec.append(get_encoder_or_decoder_block(mode="decoder", in_dims=self.out_dims_list[j],
                                                    out_dims=self.in_dims_list[j],
                                                    mlp_hidden_dims=self.in_dims_list[j] * mlp_hidden_dims_ratio, **common))
        self.encoder_list, self.decoder_list = nn.ModuleList(enc), nn.ModuleList(dec)
        k = final_conv_layer_kernel_size
        self.final_layer = nn.Sequential(   # a013:98-148
            nn.Conv2d(2, 2, kernel_size=k, padding="same", padding_mode="reflect"),
            nn.BatchNorm2d(2),
            mlp_activation_func,
            nn.Conv2d(2, 1, kernel_size=k, padding="same", padding_mode="reflect"),
        )
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh_weights())

    # ---- weight arena ---------------------------------------------------------------------------
    def _model_desc(self) -> L.ModelDesc:
        n = len(self.in_dims_list)
        if n > L.SWF_MAX_LEVELS:
            raise NotImplementedError(f"at most {L.SWF_MAX_LEVELS} levels")
        d = L.ModelDesc()
        d.levels = n
        for j in range(n):
            d.in_dims[j], d.out_dims[j] = self.in_dims_list[j], self.out_dims_list[j]
            d.head_dim[j] = math.floor(self.out_dims_list[j] * self.att_dims_per_head_ratio)
        d.heads, d.mlp_ratio = self.att_num_heads, self.mlp_hidden_dims_ratio
        d.win_h, d.win_w = self.window_size
        d.merge_h, d.merge_w = self.merging_size
        d.head_ksize = self.final_layer_conv_kernel_size
        d.precision = _precision_code(self.precision)
        if self.schedule not in ("latency", "throughput"):
            raise ValueError(f"schedule must be 'latency' or 'throughput', got {self.schedule!r}")
        d.schedule = 1 if self.schedule == "throughput" else 0
        return d

    def refresh_weights(self) -> None:
        """Drop the packed arena; it is rebuilt from the current parameters at the next forward.
        Called automatically after load_state_dict() and .to(); call it by hand after editing
        parameters in place."""
        self._arena, self._arena_key, self._packed = None, None, None
        self.weights_epoch += 1

    def _apply(self, fn, *a, **kw):
        self._arena, self._arena_key, self._packed = None, None, None
        self.weights_epoch = getattr(self, "weights_epoch", 0) + 1
        return super()._apply(fn, *a, **kw)

    def graph_key(self):
        """Everything a captured forward depends on besides the input shape: the weights epoch, the arithmetic mode and
        the addresses of the arena / packed images (None before the first forward)."""
        return (self.weights_epoch, self.precision, self.schedule, None if self._arena is None else self._arena.data_ptr(),
                None if self._packed is None else self._packed.data_ptr())

    def param_layout(self):
        """[(state_dict key, element offset, numel)] of the arena, as defined by the library."""
        lib, desc = L.lib(), self._model_desc()
        n = lib.swf_model_param_count(C.byref(desc))
        if n < 0:
            L.check(L.ERR_BAD_SHAPE)
        buf = C.create_string_buffer(512)
        off, num = C.c_int64(), C.c_int64()
        out = []
        for i in range(n):
            L.check(lib.swf_model_param_info(C.byref(desc), i, buf, 512, C.byref(off), C.byref(num)))
            out.append((buf.value.decode(), off.value, num.value))
        return out

    def _get_arena(self, device) -> Tensor:
        key = (torch.device(device).index,)
        if self._arena is None or self._arena_key != key:
            _require_elu(self.mlp_activation_func)
            lib, desc = L.lib(), self._model_desc()
            total = lib.swf_model_arena_elems(C.byref(desc))
            sd = self.state_dict()
            layout = self.param_layout()
            # packed on the device: the parameters (already there after .to()) are concatenated in layout order with the
            # library's 16-byte alignment gaps — a handful of concat kernels instead of one host round trip per tensor
            pieces, pos = [], 0
            for name, off, num in layout:
                t = sd[name]
                if t.numel() != num:
                    raise RuntimeError(f"{name}: expected {num} elements, got {t.numel()}")
                if off > pos:
                    pieces.append(torch.zeros(off - pos, dtype=torch.float32, device=device))
                pieces.append(t.detach().reshape(-1).to(device=device, dtype=torch.float32))
                pos = off + num
            if total > pos:
                pieces.append(torch.zeros(total - pos, dtype=torch.float32, device=device))
            self._arena = torch.cat(pieces)
            assert self._arena.numel() == total
            self._arena_key = key
            self._packed = None
            self.weights_epoch += 1
        return self._arena

    def _get_packed(self, arena: Tensor) -> Optional[Tensor]:
        """Kernel-layout weight images of the fused levels, derived once per arena (swf_model_pack_weights)."""
        if self._packed is None:
            lib, desc = L.lib(), self._model_desc()
            n = lib.swf_model_packed_bytes(C.byref(desc))
            buf = torch.empty(max(n, 16), dtype=torch.uint8, device=arena.device)
            if n:
                L.check(lib.swf_model_pack_weights(C.byref(desc), _ptr(arena), buf.data_ptr(), n, _stream(arena.device)))
            self._packed = buf
        return self._packed

    def load_reference_checkpoint(self, path, map_location="cpu") -> dict:
        """Load a checkpoint written by the reference's training script (a016:243-249:
        {"model_state", "optimizer_state", "scheduler_state", "current_epoch"}; loaded by a017:50-54).  The aliased
        3139-key `model_state` loads strictly.  `weights_only=True`: nothing in the file is executed."""
        state = torch.load(path, map_location=map_location, weights_only=True)
        model_state = state["model_state"] if isinstance(state, dict) and "model_state" in state else state
        self.load_state_dict(model_state, strict=True)
        return {k: v for k, v in state.items() if k != "model_state"} if isinstance(state, dict) else {}

    def _head_params(self) -> L.HeadParams:
        conv1, bn, conv2 = self.final_layer[0], self.final_layer[1], self.final_layer[3]
        ptr = lambda t: None if t is None else _ptr(t)
        return L.HeadParams(_ptr(conv1.weight), ptr(conv1.bias), _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var),
                            _ptr(conv2.weight), ptr(conv2.bias))

    def _forward_autograd(self, in_x: Tensor, in_y: Tensor) -> Tensor:
        """a013:209-230 module by module under torch.autograd (training side): every module's forward is a library
        call and its backward a library call of kernels_bwd.hip (exact fp32).  Both modes of the reference: under model.train() the
        head's BatchNorm normalises with the batch statistics and updates its running statistics (a016:137), under model.eval() it uses
        the running statistics; dropout must be 0 (the reference's configuration, A000_CONFIG.py).  The one-call fused forward
        (swf_model_forward) is not differentiable and is not used here."""
        if in_x.shape != in_y.shape or in_x.shape[1] != self.in_dims_list[0]:
            raise ValueError(f"expected two (B,{self.in_dims_list[0]},H,W) tensors, got {tuple(in_x.shape)} and {tuple(in_y.shape)}")
        self.u_net_intermediate_result_recorder.delete_all()
        if self._arena is not None:
            # a differentiable forward is a training step: an optimizer is about to change the parameters in place, which nothing
            # here could notice — drop the fused forward's weight arena now, the next no-grad forward (a016:202) rebuilds it
            self.refresh_weights()
        x, y = in_x, in_y
        if not x.requires_grad:      # the module-level autograd Functions key on their inputs
            x, y = x.detach().requires_grad_(True), y.detach().requires_grad_(True)
        n = len(self.in_dims_list)
        for j, stage in enumerate(self.encoder_list):          # a013:215-220
            for mod in stage:
                x, y = mod(x, y)
            if j < n - 1:
                self.u_net_intermediate_result_recorder.record((x, y))
        for j, stage in enumerate(self.decoder_list):          # a013:221-227
            if j > 0:
                hx, hy = self.u_net_intermediate_result_recorder.read()
                x, y = _AddFunction.apply(x, hx), _AddFunction.apply(y, hy)
            for mod in stage:
                x, y = mod(x, y)
        return self.do_final_layer(x, y)

    def do_final_layer(self, x: Tensor, y: Tensor) -> Tensor:
        """a013:126-152 on its own: cat -> conv -> BatchNorm2d (batch statistics under train(), running statistics under eval()) ->
        ELU -> conv, differentiable."""
        conv1, bn, conv2 = self.final_layer[0], self.final_layer[1], self.final_layer[3]
        return _HeadFunction.apply(self, x, y, conv1.weight, conv1.bias, bn.weight, bn.bias, conv2.weight, conv2.bias)

    # ---- forward ----------------------------------------------------------------------------------
    def forward(self, in_x: Tensor, in_y: Tensor) -> Tensor:
        """a013:209-230 as one C call.  The FIRST forward of a model (and every forward after a failed input check)
        reads the per-block identical-streams flags back to the host (a005:98-118), so it synchronises and cannot be
        captured into a hipGraph: run one eager forward first (shard.ShardedFusion does).
        With torch.autograd recording (grad mode on and an input or parameter requiring grad) the forward runs module by module
        instead and is differentiable (_forward_autograd)."""
        if _wants_grad(self, in_x, in_y):
            return self._forward_autograd(in_x, in_y)
        _check_forward_only(self, in_x, in_y)
        if self.training:
            raise RuntimeError("MyModel's HIP path is the eval() forward (BatchNorm running statistics, "
                               "a013:133); call model.eval()")
        if in_x.shape != in_y.shape or in_x.shape[1] != self.in_dims_list[0]:
            raise ValueError(f"expected two (B,{self.in_dims_list[0]},H,W) tensors, got {tuple(in_x.shape)} and {tuple(in_y.shape)}")
        self.u_net_intermediate_result_recorder.delete_all()
        b, _, h, w = in_x.shape
        x, y = in_x.contiguous(), in_y.contiguous()
        arena = self._get_arena(x.device)
        out = torch.empty((b, 1, h, w), dtype=torch.float32, device=x.device)
        lib, desc = L.lib(), self._model_desc()
        need = lib.swf_model_workspace_bytes(C.byref(desc), b, h, w)
        ws, wsn = _workspace(need, x.device)
        packed = self._get_packed(arena) if desc.precision == L.PREC_FAST else None
        if self.input_compatibility_with_cross_option is None:
            # First forward: the reference checks, in front of every cross-attention block, that its two input streams
            # are not identical everywhere, and otherwise prints and calls exit() (a005:98-118).  The library records
            # the same test per cross block on the device; here it raises ValueError (documented deviation).
            flags = torch.ones(4 * len(self.in_dims_list), dtype=torch.int32, device=x.device)
            L.check(lib.swf_model_forward_checked(C.byref(desc), _ptr(arena), packed.data_ptr() if packed is not None else None,
                                                  _ptr(x), _ptr(y), _ptr(out), b, h, w, ws, wsn, flags.data_ptr(),
                                                  _stream(x.device)))
            same = flags.cpu().nonzero().flatten().tolist()
            if same:
                # stays None: the next forward runs the check again (the reference would have exited here, a005:118)
                n = len(self.in_dims_list)
                where = [f"{'encoder' if i // 2 < n else 'decoder'}_list.{(i // 2) % n} cross block {i % 2}" for i in same]
                raise ValueError("inputs are incompatible with the cross_attr option: cross attention received identical "
                                 f"x and y ({', '.join(where)}); the reference calls exit() here (a005:111-118)")
            self.input_compatibility_with_cross_option = True
            return out
        if desc.precision == L.PREC_FAST:
            L.check(lib.swf_model_forward_packed(C.byref(desc), _ptr(arena), packed.data_ptr(), _ptr(x), _ptr(y), _ptr(out),
                                                 b, h, w, ws, wsn, _stream(x.device)))
        else:
            L.check(lib.swf_model_forward(C.byref(desc), _ptr(arena), _ptr(x), _ptr(y), _ptr(out), b, h, w, ws, wsn,
                                          _stream(x.device)))
        return out
