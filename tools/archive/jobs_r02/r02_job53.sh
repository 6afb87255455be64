#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job53
mkdir -p $OUT
cd $ROOT
for m in 2 10 6 14 8; do
  echo "== RK_PC_MID_MASK=$m" | tee -a $OUT/mask.txt
  RK_PC_MID_MASK=$m python3 tools/shard_sim.py 4000000 0,0 2>&1 | grep "N=8\|N=4" | tee -a $OUT/mask.txt
  RK_PC_MID_MASK=$m python3 tools/size_sweep.py 3.5e5,5e5 2>&1 | grep -v amdgpu | tee -a $OUT/mask.txt
done
