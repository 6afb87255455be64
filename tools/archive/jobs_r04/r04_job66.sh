#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2 3; do
for v in 30000 60000; do
  echo "RK_PLAN_REV_MAX_GROUPS=$v $(RK_PLAN_REV_MAX_GROUPS=$v timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep 'N=2 work')"
done; done
