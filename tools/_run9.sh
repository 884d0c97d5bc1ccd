cd $GRAFT_REPO_ROOT
export SWF_LIB_PATH=$PWD/swin_unet_image_fusion_amd/libswf_af16.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "(basic_block and (w8 or hid4)) or block_pair or model_golden or nonsquare or ragged or checkpoint" > gpurun_out/t_r02g.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/t_r02g.log
cp gpurun_out/parity.json gpurun_out/parity_af16.json
sed -i 's/for v in "" swin_unet_image_fusion_amd\/libswf_w4.so; do/for v in "" swin_unet_image_fusion_amd\/libswf_af16.so; do/' tools/_run8.sh
bash tools/_run8.sh
