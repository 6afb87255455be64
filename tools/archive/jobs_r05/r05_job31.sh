#!/bin/bash
# Where the one-launch kernel (k_list_any, 5 waves per SIMD) hands over to the four class kernels (8 / 8 / 6 / 6) on round 5's kernels:
# RK_PLAN_MAX_GROUPS = 30000 (default) / 20000 / 10000 on the shards of the 4M tree (27k / 13.4k nodes) and on whole trees of 0.5M-1.5M.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job31
mkdir -p $O
for g in 30000 20000 10000; do
  echo "== RK_PLAN_MAX_GROUPS=$g RK_PLAN_REV_MAX_GROUPS=$g"
  RK_PLAN_MAX_GROUPS=$g RK_PLAN_REV_MAX_GROUPS=$g timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work\|full" | tee -a $O/shards_$g.txt
  RK_PLAN_MAX_GROUPS=$g RK_PLAN_REV_MAX_GROUPS=$g timeout 600 python3 tools/pc_ring_probe.py 500000,750000,1000000,1500000 2>&1 | tail -1 | tee -a $O/probe_$g.txt
done
