#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job41
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for cfg in "RK_PLAN=1" "RK_PLAN=2 RK_PLAN_TAIL=0.25" "RK_PLAN=2 RK_PLAN_TAIL=0.4" "RK_PLAN=2 RK_PLAN_BUCKETS=4" "RK_PLAN=2 RK_PLAN_BUCKETS=8"; do
    echo -n "$cfg: " | tee -a $OUT/ab.txt
    env $cfg python3 tools/step_gap.py 2>&1 | grep "ms per call" | sed 's/.*back to back/b2b/' | tee -a $OUT/ab.txt
  done
done
