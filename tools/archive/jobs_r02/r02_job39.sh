#!/bin/bash
# 4M: Morton order with the lightest critical nodes moved to the end of the dispatch order (experiment).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job39
mkdir -p $OUT
cd $ROOT
python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/tail.txt
for f in 0.03 0.06 0.12 0.25; do
  echo "RK_PLAN=2 RK_PLAN_TAIL=$f" | tee -a $OUT/tail.txt
  RK_PLAN=2 RK_PLAN_TAIL=$f python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/tail.txt
done
echo "RK_XCD_MODE=0" | tee -a $OUT/tail.txt
RK_XCD_MODE=0 python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/tail.txt
