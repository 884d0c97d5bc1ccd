# PMC passes over one BasicBlock launch loop (tools/profile_block.py), one rocprofv3 --pmc run per counter group.
#   bash tools/pmc_block.sh <level> <outdir-under-gpurun_out> [groups...]      (PB_EXTRA="--decoder 1" adds profile_block.py arguments)
set -e
LEVEL=${1:-1}
O=gpurun_out/${2:-pmc_l$LEVEL}
shift 2 || true
GROUPS_="${@:-g1 g2 g3 g4 g5}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$n -o p -- python3 tools/profile_block.py --level $LEVEL --iters 10 $PB_EXTRA > $O.$n.log 2>&1; }
for g in $GROUPS_; do
  case $g in
    g1) run g1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ;;
    g2) run g2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT ;;
    g3) run g3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES ;;
    g4) run g4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH ;;
    g5) run g5 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || echo "g5 failed" ;;
  esac
done
