#!/bin/bash
# Same question upwards: heavy-first plan + k_list_any up to 45000 / 60000 critical nodes instead of 30000.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for g in 30000 45000 60000; do
  echo "== RK_PLAN_MAX_GROUPS=$g"
  RK_PLAN_MAX_GROUPS=$g RK_PLAN_REV_MAX_GROUPS=$g timeout 600 python3 tools/shard_sim.py 2>&1 | grep "N=2 work\|full"
  RK_PLAN_MAX_GROUPS=$g RK_PLAN_REV_MAX_GROUPS=$g timeout 600 python3 tools/pc_ring_probe.py 1250000,1500000,1750000,2000000 2>&1 | tail -1
done
