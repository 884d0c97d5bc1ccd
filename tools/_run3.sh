cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "basic_block or block_pair or model_golden or full_size or nonsquare or ragged or prepack" > gpurun_out/t_r02c.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/t_r02c.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r02c.json 2> gpurun_out/bench_r02c.err; rc=$?; echo "bench rc=$rc"; cat gpurun_out/bench_r02c.json; tail -3 gpurun_out/bench_r02c.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
python tools/profile_block.py --level 0 --iters 30 ; python tools/profile_block.py --level 0 --iters 30 --decoder 1; python tools/profile_block.py --level 0 --iters 30 --shift 0 --cross 0
