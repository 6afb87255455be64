#!/bin/bash
# VALU instruction count and lane utilisation of the 100k launch (is it issue-bound like the 4M one?), plus the failing/new tests again.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r04_pmc100k_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload plummer100k_f32 --no-pageable-leg > $ROOT/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES
run g GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VALU
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A30 "k_pc_any\|k_super" $OUT/summary.txt | head -80

