# Per-level PMC of one BasicBlock launch loop (tools/profile_block.py), encoder widths, levels 0..4: instruction mix, MFMA / VALU busy,
# LDS conflicts, clock (GRBM_GUI_ACTIVE), and HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes, level 0 only).
#   bash tools/pmc_levels.sh <outdir-under-gpurun_out>
#   LEVELS="2 3" SCHED=throughput bash tools/pmc_levels.sh <outdir>     (the kernels of the throughput schedule; no traffic passes)
set -e
O=gpurun_out/${1:-pmc_levels}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
SCHED=${SCHED:-latency}
run() { lvl=$1; n=$2; shift 2; rocprofv3 --pmc "$@" --output-format csv -d $O/l$lvl/$n -o p -- python3 tools/profile_block.py --level $lvl --iters 10 --schedule $SCHED > $O/l$lvl.$n.log 2>&1 || echo "level $lvl group $n failed"; }
for lvl in ${LEVELS:-0 1 2 3 4}; do
  mkdir -p $O/l$lvl
  run $lvl g1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  run $lvl g2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT
  run $lvl g3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE
  run $lvl g4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
  echo "level $lvl done"
done
if [ -n "$LEVELS" ]; then exit 0; fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/traffic/fetch -o p -- python3 tools/profile_block.py --level 0 --iters 10 > $O/traffic.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/traffic/write -o p -- python3 tools/profile_block.py --level 0 --iters 10 > $O/traffic.write.log 2>&1
echo "traffic done"
