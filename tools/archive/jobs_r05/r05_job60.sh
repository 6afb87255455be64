#!/bin/bash
# Supergroup size of the pre-pass (RK_SUPER_K) at small and medium launches, pre-pass recomputed every call (RK_SUPER_CACHE=0):
# kernel ms (pre-pass + traversal), result hashes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export RK_SUPER_CACHE=0
for rep in 1 2; do
for k in 16 8 12 24 32; do
  echo -n "K=$k "; RK_SUPER_K=$k timeout 600 python3 tools/pc_ring_probe.py 100000,350000,1000000,2000000,4000000 2>&1 | tail -1
done
done
for k in 16 8 32; do
  echo "shards K=$k"; RK_SUPER_K=$k timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep "work\|full"
done
