cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t_r02e.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/t_r02e.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r02e.json 2> gpurun_out/bench_r02e.err; rc=$?; echo "bench rc=$rc"; cat gpurun_out/bench_r02e.json
