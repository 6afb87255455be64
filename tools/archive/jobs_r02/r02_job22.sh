#!/bin/bash
# Wave issue priorities by position in the heavy-first lists (RK_PRIO): strong-scaling shards and small sizes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job22
mkdir -p $OUT
cd $ROOT
for pr in 0 1 2 3; do
  echo "== RK_PRIO=$pr" | tee -a $OUT/sweep.txt
  RK_PRIO=$pr timeout 600 python3 tools/size_sweep.py 1e5,3.5e5,5e5,1e6,2e6 >> $OUT/sweep.txt 2>&1
  RK_PRIO=$pr timeout 600 python3 tools/shard_sim.py 4000000 0,0 2>&1 | grep -v particles >> $OUT/sweep.txt
done
cat $OUT/sweep.txt
