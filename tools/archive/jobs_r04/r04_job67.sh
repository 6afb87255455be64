#!/bin/bash
# N = 4 shards of the 4M tree (27k nodes each): heavy-first one-launch plan (default, limit 30000) against the light-tail plan
# on the class kernels (limit 20000), and N = 8 shards (13.4k nodes) with limit 10000.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for v in 30000 20000 10000; do
  echo "RK_PLAN_MAX_GROUPS=$v $(RK_PLAN_MAX_GROUPS=$v RK_PLAN_REV_MAX_GROUPS=$v timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep 'work' | sed -n 2,3p | tr '\n' ' ')"
done; done
