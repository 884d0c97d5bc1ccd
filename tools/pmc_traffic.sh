# HBM bytes per launch of the level-<L> block kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes.
#   bash tools/pmc_traffic.sh <level> <outdir-under-gpurun_out>
set -e
LEVEL=${1:-0}
O=gpurun_out/${2:-traffic_l$LEVEL}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 tools/profile_block.py --level $LEVEL --iters 10 > $O.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 tools/profile_block.py --level $LEVEL --iters 10 > $O.write.log 2>&1
