"""Phase stamps of workgroup 0 / wave 0 of window96x8_kernel (diagnostic library built with -DW96_PROBE, loaded through SWF_LIB_PATH).

    python -c "import __graft_entry__ as g; g.build()"            # objects under swin_unet_image_fusion_amd/csrc/build/
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DW96_PROBE -c swin_unet_image_fusion_amd/csrc/kernels_win96.hip -o /tmp/w96p.o
    hipcc -shared -fPIC --offload-arch=gfx950 -o swin_unet_image_fusion_amd/libswf_probe96.so \
          $(ls swin_unet_image_fusion_amd/csrc/build/*.o | grep -v kernels_win96) /tmp/w96p.o
    [W96_BATCH=32] SWF_LIB_PATH=$PWD/swin_unet_image_fusion_amd/libswf_probe96.so python tools/w96_probe.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from swin_unet_image_fusion_amd import CONFIGS, MyModel, _lib as L, load_recipe_into
from swin_unet_image_fusion_amd.modules import _ptr, _stream
torch.set_grad_enabled(False)
cfg = CONFIGS["win8"]
model = MyModel(**cfg.model_kwargs(nn.ELU(inplace=True))).eval(); load_recipe_into(model, seed=0); model.to("cuda:0")
lib = L.lib()
lvl, side = 2, 32
BATCH = int(os.environ.get("W96_BATCH", "16"))   # 32: two windows per workgroup (the second runs with a warm instruction cache)
names = ["LN1 done", "Q/K/V done", "images barrier", "attention", "projection", "exchange 1", "LN2", "MLP loop", "before store", "stored"]
for dec, shift, cross in ((0, 1, 1), (1, 1, 1)):
    stage = model.decoder_list[4 - lvl][0] if dec else model.encoder_list[lvl][3]
    grp = stage.cross_att_block if cross else stage.self_att_block
    blk = grp.shifted_window_block if shift else grp.normal_window_block
    c = blk.in_out_dims
    x = torch.randn(BATCH, side, side, c, device="cuda:0"); y = torch.randn(BATCH, side, side, c, device="cuda:0"); ox, oy = torch.empty_like(x), torch.empty_like(y)
    desc = blk._desc("fast"); px, py = blk._stream_params("x"), blk._stream_params("y")
    n = lib.swf_basic_block_packed_bytes(C.byref(desc)); packed = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    st = _stream(x.device)
    L.check(lib.swf_basic_block_pack(C.byref(desc), C.byref(px), C.byref(py), packed.data_ptr(), n, st))
    run = lambda: L.check(lib.swf_basic_block_fwd_packed(C.byref(desc), packed.data_ptr(), _ptr(x), _ptr(y), _ptr(ox), _ptr(oy), BATCH, side, side, st))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    raw = C.CDLL(L.LIB_PATH)
    assert raw.swf_w96_probe_read(buf) == 0
    t = [(buf[i] - buf[0]) * 0.01 for i in range(11)]
    if BATCH > 16:
        print(f"  first window {(buf[12] - buf[11]) * 0.01:.2f} us, second window {(buf[10] - buf[12]) * 0.01:.2f} us (stamps below are the second window's, from kernel entry)")
    print(f"level 2 {'dec' if dec else 'enc'} hidden {blk.mlp_hidden if hasattr(blk, 'mlp_hidden') else '?'}: {e0.elapsed_time(e1) * 50:.1f} us per launch | " +
          " | ".join(f"{names[i]} {t[i + 1]:.2f}" for i in range(10)))
