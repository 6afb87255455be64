#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job42; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
# a parent that holds a GPU context and memory meanwhile, like the pytest process does
python3 -c "
import torch, time
a = torch.zeros(1 << 28, device='cuda'); torch.cuda.synchronize(); time.sleep(1500)" &
PARENT=$!
sleep 8
fails=0
for i in $(seq 1 220); do
  for plan in 0 1; do
    RK_PLAN=$plan timeout 120 python3 tools/first_call_loop.py > $OUT/child.log 2>&1; rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); cp $OUT/child.log $OUT/fail_${i}_$plan.log; echo "iteration $i plan $plan rc=$rc"; grep -v "^  File" $OUT/child.log | tail -40; fi
  done
  [ $fails -ge 3 ] && break
done
echo "done: $fails failures in $i iterations"
kill $PARENT
