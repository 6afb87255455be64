#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job54
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for cfg in "RK_PLAN=1" "RK_PLAN_WEIGHT=size" "RK_PLAN=0"; do
    echo -n "$cfg: " | tee -a $OUT/ab.txt
    env $cfg python3 tools/step_gap.py 2>&1 | grep "ms per call" | sed 's/.*back to back/b2b/' | tee -a $OUT/ab.txt
  done
done
for cfg in "RK_PLAN=1" "RK_PLAN_WEIGHT=size" "RK_PLAN=0"; do
  echo -n "$cfg: " | tee -a $OUT/ab.txt
  env $cfg python3 tools/size_sweep.py 1e5,5e5,1e6,2e6 2>&1 | grep -v amdgpu | tr '\n' ';' | tee -a $OUT/ab.txt; echo | tee -a $OUT/ab.txt
done
