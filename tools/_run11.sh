cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide or model_golden or full_size_batch or nonsquare" > gpurun_out/t_r02h.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/t_r02h.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r02h.json 2> gpurun_out/bench_r02h.err; rc=$?; echo "bench rc=$rc"; cut -c1-200 gpurun_out/bench_r02h.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tr_r02h -o t -- python3 tools/trace_forward.py run > gpurun_out/tr_r02h.log 2>&1
python3 tools/trace_forward.py report gpurun_out/tr_r02h/*/t_results.db > gpurun_out/r02h_forward_timeline.txt 2>&1 || python3 tools/trace_forward.py report $(find gpurun_out/tr_r02h -name "*.db" | head -1) > gpurun_out/r02h_forward_timeline.txt 2>&1
tail -3 gpurun_out/r02h_forward_timeline.txt
