cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "model_golden or full_size or nonsquare or ragged or prepack or packed or checkpoint or sharded" > gpurun_out/t_r02j.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/t_r02j.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-180
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tr_r02j -o t -- python3 tools/trace_forward.py run > gpurun_out/tr_r02j.log 2>&1
python3 tools/trace_forward.py report $(find gpurun_out/tr_r02j -name "*.db" | head -1) > gpurun_out/r02j_forward_timeline.txt 2>&1
grep -E "window48|window24" gpurun_out/r02j_forward_timeline.txt | head -20; tail -1 gpurun_out/r02j_forward_timeline.txt
