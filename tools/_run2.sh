cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t_r02b.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/t_r02b.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; rc=$?; echo "bench rc=$rc"; cat gpurun_out/bench_r02b.json; tail -3 gpurun_out/bench_r02b.err
