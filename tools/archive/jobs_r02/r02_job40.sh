#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job40
mkdir -p $OUT
cd $ROOT
python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/tail.txt
for f in 0.4 0.6; do
  echo "RK_PLAN=2 RK_PLAN_TAIL=$f" | tee -a $OUT/tail.txt
  RK_PLAN=2 RK_PLAN_TAIL=$f python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/tail.txt
done
for b in 2 3 4 8 16 64; do
  echo "RK_PLAN=2 RK_PLAN_BUCKETS=$b" | tee -a $OUT/tail.txt
  RK_PLAN=2 RK_PLAN_BUCKETS=$b python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/tail.txt
done
