cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/t_r02a.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/t_r02a.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; rc=$?; echo "bench rc=$rc"; cat gpurun_out/bench_r02a.json
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
bash tools/pmc_block.sh 0 pmc_r02a_l0enc g1 g2 g3 g4 && PB_EXTRA="--decoder 1" bash tools/pmc_block.sh 0 pmc_r02a_l0dec g1 g2 g3
