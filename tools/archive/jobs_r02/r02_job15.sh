#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export PYTHONUNBUFFERED=1
export RK_PLAN_MAX_GROUPS=200000 RK_PC_R2_BELOW=200000
for frac in 0 0.15 0.3 0.6 1; do
  echo "== RK_PC_FRAC=$frac"
  for n in 1000000 2000000 4000000; do
  RK_PC_FRAC=$frac timeout 300 python3 bench.py --no-cpu-baseline --nparts $n --steps 30 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  n $n: value', d['value'], 'kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'])"
  done
  RK_PC_FRAC=$frac timeout 600 python3 tools/shard_sim.py 4000000 0,0 2>&1 | grep -v amdgpu | grep "N=8\|N=4" | head -2
done
