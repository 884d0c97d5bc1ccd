cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide and C192" 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-180
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tr_r02i -o t -- python3 tools/trace_forward.py run > gpurun_out/tr_r02i.log 2>&1
python3 tools/trace_forward.py report $(find gpurun_out/tr_r02i -name "*.db" | head -1) > gpurun_out/r02i_forward_timeline.txt 2>&1
grep -E "qkv_attn" gpurun_out/r02i_forward_timeline.txt | head -4; tail -1 gpurun_out/r02i_forward_timeline.txt
