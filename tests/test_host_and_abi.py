"""CPU-only checks: the C-ABI library loads and exports every symbol include/swinfuse.h declares,
argument validation returns the documented status codes without touching a GPU, and the host-side
module mirror reproduces the reference's state_dict key set / shapes / aliasing (golden key tables
captured from the real reference)."""
import ctypes as C
import json
import os
import re

import pytest
import torch
from torch import nn

import __graft_entry__ as entry
from swin_unet_image_fusion_amd import CONFIGS, MyModel, SelfAndCrossBlockPair, WindowAttention, _lib as L
from swin_unet_image_fusion_amd.config import alias_groups_from_tensors, load_recipe_into, make_state_arrays
from tests import golden_util as G

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    entry.build()


def _header_functions():
    text = open(os.path.join(REPO, "include", "swinfuse.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(swf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 25
    handle = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in swinfuse.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(L.SIGNATURES) == names


def test_version_and_status_strings():
    lib = L.lib()
    assert lib.swf_version() == 1
    assert lib.swf_status_string(0) == b"ok"
    assert b"pad" in lib.swf_status_string(L.ERR_PAD)


def test_argument_validation_without_gpu():
    lib = L.lib()
    desc = L.AttnDesc(8, 2, 4, 4, 4, 0)
    prm = L.AttnParams()
    # NULL tensors
    assert lib.swf_window_attention_fwd(C.byref(desc), C.byref(prm), None, None, None, None, None, 1, 8, 8, None, 0, None) == L.ERR_NULL
    # map not a multiple of the window -> BAD_SHAPE (einops error in the reference)
    assert lib.swf_window_attention_fwd(C.byref(desc), C.byref(prm), 1, 1, 1, None, 1, 1, 9, 8, None, 0, None) == L.ERR_BAD_SHAPE
    with pytest.raises(ValueError):
        L.check(L.ERR_BAD_SHAPE)
    # reflect pad >= dim -> ERR_PAD -> RuntimeError (a006:128)
    hm, wm, ho, wo = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    st = lib.swf_merge_out_shape(8, 8, 2, 2, 8, 8, C.byref(hm), C.byref(wm), C.byref(ho), C.byref(wo))
    assert st == L.ERR_PAD
    with pytest.raises(RuntimeError):
        L.check(st)
    assert lib.swf_merge_out_shape(200, 200, 2, 2, 7, 7, C.byref(hm), C.byref(wm), C.byref(ho), C.byref(wo)) == 0
    assert (hm.value, wm.value, ho.value, wo.value) == (100, 100, 105, 105)
    assert lib.swf_merge_out_shape(5, 4, 2, 2, 1, 1, C.byref(hm), C.byref(wm), C.byref(ho), C.byref(wo)) == 0
    assert (hm.value, wm.value) == (3, 2)


@pytest.mark.parametrize("cfg_name", ["win8", "win7", "tiny", "tiny7", "win8_4stage", "win16"])
def test_state_dict_matches_reference_key_table(cfg_name):
    with open(os.path.join(G.GOLDEN, f"state_keys_{cfg_name}.json")) as f:
        ref = json.load(f)
    m = MyModel(**CONFIGS[cfg_name].model_kwargs(nn.ELU(inplace=True)))
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref["shapes"].keys())
    for k, shape in ref["shapes"].items():
        assert list(sd[k].shape) == shape, k
    alias = {k: a for k, a in alias_groups_from_tensors(sd).items() if a != k}
    assert alias == ref["alias_of"]


def test_arena_layout_covers_every_parameter_once():
    m = MyModel(**CONFIGS["win8"].model_kwargs(nn.ELU(inplace=True)))
    layout = m.param_layout()
    sd = m.state_dict()
    names = [n for n, _, _ in layout]
    assert len(names) == len(set(names))
    canon = set(alias_groups_from_tensors(sd).values())
    skipped = {k for k in canon if k.endswith("buffer_to_show_device") or k.endswith("num_batches_tracked")}
    assert set(names) == canon - skipped
    end = 0
    for n, off, num in layout:
        assert off >= end and off % 4 == 0 and sd[n].numel() == num
        end = off + num
    assert end <= L.lib().swf_model_arena_elems(C.byref(m._model_desc()))
    # 33 150 453 learnable parameters; the arena also carries BatchNorm running_mean/var (2 + 2)
    assert sum(num for _, _, num in layout) == 33150453 + 4

def test_recipe_is_deterministic_and_module_independent():
    """The weight recipe depends only on the key table, so the reference model (fixtures) and this
    package's mirror get identical tensors."""
    m = MyModel(**CONFIGS["tiny"].model_kwargs(nn.ELU(inplace=True)))
    load_recipe_into(m, seed=0, flavor="stress")
    with open(os.path.join(G.GOLDEN, "state_keys_tiny.json")) as f:
        ref = json.load(f)
    arrays = make_state_arrays({k: tuple(v) for k, v in ref["shapes"].items()}, ref["alias_of"], seed=0, flavor="stress")
    sd = m.state_dict()
    for k, a in arrays.items():
        assert torch.equal(sd[k], torch.from_numpy(a).to(sd[k].dtype)), k


def test_forward_refuses_cpu_tensors_and_grad():
    wa = WindowAttention(8, 2, 4, (4, 4), False, False, True, 0.0, 0.0).eval()
    x = torch.zeros(1, 8, 8, 8)
    with torch.no_grad(), pytest.raises(RuntimeError):
        wa(x, x, x)            # CPU tensor: no fallback, loud failure
    with pytest.raises(RuntimeError):
        wa(x.requires_grad_(), x, x)
    m = MyModel(**CONFIGS["tiny"].model_kwargs(nn.ELU(inplace=True)))   # training mode
    with torch.no_grad(), pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 16, 16), torch.ones(1, 1, 16, 16))


def test_product_path_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/ (checker, never product)."""
    pkg = os.path.join(REPO, "swin_unet_image_fusion_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+oracle)|import_module\([^)]*oracle|__import__\([^)]*oracle", re.M)
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                assert not pat.search(open(os.path.join(root, fn)).read()), f"{fn} imports the oracle"


def test_reference_format_checkpoint_roundtrip(tmp_path):
    """A checkpoint in the reference's on-disk format (a016:243-249) loads strictly through weights_only=True."""
    a = MyModel(**CONFIGS["tiny"].model_kwargs(nn.ELU(inplace=True)))
    load_recipe_into(a, seed=4, flavor="stress")
    path = tmp_path / "ckpt.pth"
    torch.save({"model_state": a.state_dict(), "optimizer_state": {}, "scheduler_state": {}, "current_epoch": 7}, path)
    b = MyModel(**CONFIGS["tiny"].model_kwargs(nn.ELU(inplace=True)))
    rest = b.load_reference_checkpoint(str(path))
    assert rest["current_epoch"] == 7
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    # aliases stay aliases after loading (one storage behind stage_1.other_module.* and auto_path_win_att.*)
    sd = b.state_dict()
    k1 = "encoder_list.0.3.self_att_block.normal_window_block.auto_path_win_att.window_attention_x.q_for_heads.weight"
    k2 = "encoder_list.0.3.self_att_block.normal_window_block.stage_1.other_module.window_attention_x.q_for_heads.weight"
    assert sd[k1].data_ptr() == sd[k2].data_ptr()
