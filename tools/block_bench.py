"""Time the fused BasicBlock launch of one level with pre-packed weights (swf_basic_block_fwd_packed): best of 5 x 20 launches.
    python tools/block_bench.py <level>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from swin_unet_image_fusion_amd import CONFIGS, MyModel, _lib as L, load_recipe_into
from swin_unet_image_fusion_amd.modules import _ptr, _stream
torch.set_grad_enabled(False)
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = CONFIGS["win8"]
model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval(); load_recipe_into(model, seed=0); model.to("cuda:0")
lib = L.lib()
side = 256 >> (lvl + 1)
for dec, shift, cross in ((0, 1, 1), (0, 0, 0), (1, 1, 1), (1, 0, 0)):
    stage = model.decoder_list[4 - lvl][0] if dec else model.encoder_list[lvl][3]
    grp = stage.cross_att_block if cross else stage.self_att_block
    blk = grp.shifted_window_block if shift else grp.normal_window_block
    c = blk.in_out_dims
    x = torch.randn(16, side, side, c, device="cuda:0"); y = torch.randn(16, side, side, c, device="cuda:0"); ox, oy = torch.empty_like(x), torch.empty_like(y)
    desc = blk._desc("fast"); px, py = blk._stream_params("x"), blk._stream_params("y")
    n = lib.swf_basic_block_packed_bytes(C.byref(desc)); packed = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    st = _stream(x.device)
    L.check(lib.swf_basic_block_pack(C.byref(desc), C.byref(px), C.byref(py), packed.data_ptr(), n, st))
    run = lambda: L.check(lib.swf_basic_block_fwd_packed(C.byref(desc), packed.data_ptr(), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy), 16, side, side, st))
    for _ in range(5): run()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"level {lvl} C={c} hid={blk.mlp_hidden_dims} dec={dec} shift={shift} cross={cross}: {best:.1f} us per launch (best of 5x20)", "finite", bool(torch.isfinite(ox).all()))
