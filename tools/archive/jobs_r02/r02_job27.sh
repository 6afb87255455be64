#!/bin/bash
# Launch-plan granularity (RK_PLAN_K, RK_PLAN_KEY), block mapping (RK_PLAN_XCD) and wave priorities (RK_PRIO) on small launches.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job27
mkdir -p $OUT
cd $ROOT
for cfg in "0 0 0 0" "1 0 0 0" "1 0 0 1" "1 0 2 0" "1 0 2 1" "4 0 0 0" "4 0 0 1" "16 1 0 0" "16 1 0 1" "4 1 2 1"; do
  set -- $cfg
  echo "== RK_PLAN_K=$1 RK_PLAN_KEY=$2 RK_PLAN_XCD=$3 RK_PRIO=$4" >> $OUT/sweep.txt
  RK_PLAN_K=$1 RK_PLAN_KEY=$2 RK_PLAN_XCD=$3 RK_PRIO=$4 timeout 600 python3 tools/size_sweep.py 1e5,3.5e5,1e6 >> $OUT/sweep.txt 2>&1
  RK_PLAN_K=$1 RK_PLAN_KEY=$2 RK_PLAN_XCD=$3 RK_PRIO=$4 timeout 600 python3 tools/shard_sim.py 4000000 0,0 2>&1 | grep "N=8\|N=4" >> $OUT/sweep.txt
done
grep -v amdgpu.ids $OUT/sweep.txt
