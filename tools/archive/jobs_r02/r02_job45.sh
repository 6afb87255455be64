#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job45
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for cfg in "RK_PLAN_TAIL_XCD=0" "RK_PLAN_TAIL_XCD=5" "RK_PLAN_TAIL_XCD=1" "RK_PLAN=0"; do
    echo -n "$cfg: " | tee -a $OUT/ab.txt
    env $cfg python3 tools/step_gap.py 2>&1 | grep "ms per call" | sed 's/.*back to back/b2b/' | tee -a $OUT/ab.txt
  done
done
