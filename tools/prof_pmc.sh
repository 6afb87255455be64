#!/bin/bash
# PMC passes for the traversal kernel (run on the GPU box through gpurun). Counters are collected in
# separate passes with --pmc only (no tracing domains), as the pool requires.
# usage: tools/prof_pmc.sh <outdir> [bench args...]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pageable-leg $BENCH_ARGS > $ROOT/$OUT/$name.log 2>&1
}
BENCH_ARGS="${BENCH_ARGS:-}"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE GRBM_GUI_ACTIVE
