#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export PYTHONUNBUFFERED=1
for mask in 14 10 8 2; do
  echo "== RK_PC_MASK=$mask"; RK_PC_MASK=$mask timeout 600 python3 tools/pc_check.py 100000 350000 1000000 2>&1 | grep -v amdgpu | grep "q=0"
  RK_PC_MASK=$mask timeout 600 python3 tools/shard_sim.py 4000000 3,3 2>&1 | grep -v amdgpu | grep "full\|N=8\|N=4" | head -3
done
