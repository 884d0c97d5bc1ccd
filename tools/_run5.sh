cd $GRAFT_REPO_ROOT
for v in FOLD_MAX_0 MED3_0; do
  echo "=== $v"
  SWF_LIB_PATH=$PWD/swin_unet_image_fusion_amd/libswf_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_basic_block and (w8 or hid4) and fast" 2>&1 | tail -4
  SWF_LIB_PATH=$PWD/swin_unet_image_fusion_amd/libswf_$v.so python tools/profile_block.py --level 0 --iters 30 --shift 0 --cross 0
done
