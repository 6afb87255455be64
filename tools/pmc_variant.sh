#!/bin/bash
# PMC passes (separate --pmc runs, no tracing) of one kernel variant: tools/pmc_variant.sh <outdir> <n> <variant> [reps]
set -u
OUT=$1; N=$2; V=$3; REPS=${4:-6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  RK_SERIAL_CLASSES=1 RK_GRAPH=0 timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/tools/run_variant.py $N $V $REPS > $ROOT/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
run tcc3 WRITE_SIZE GRBM_GUI_ACTIVE
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 $ROOT/tools/pmc_summary.py $ROOT/$OUT > $ROOT/$OUT/summary.txt 2>&1
find $ROOT/$OUT -name "*.csv" -delete
