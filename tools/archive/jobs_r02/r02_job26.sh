#!/bin/bash
# Stream priorities for the class kernels (experiment).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job26
mkdir -p $OUT
cd $ROOT
for cfg in "G1:" "G0:" "G0:-1,0,0,0,0" "G0:-1,0,-1,0,0" "G0:0,-1,-1,0,0" "G1:-1,0,-1,0,0" "G0:-1,-1,-1,0,0"; do
  g=${cfg%%:*}; pr=${cfg#*:}
  echo "== RK_GRAPH=${g#G} RK_STREAM_PRIO=$pr" | tee -a $OUT/sweep.txt
  if [ -n "$pr" ]; then export RK_STREAM_PRIO=$pr; else unset RK_STREAM_PRIO; fi
  RK_GRAPH=${g#G} timeout 600 python3 tools/size_sweep.py 1e5,5e5,1e6,4e6 >> $OUT/sweep.txt 2>&1
  RK_GRAPH=${g#G} timeout 600 python3 tools/shard_sim.py 4000000 0,0 2>&1 | grep "N=8" >> $OUT/sweep.txt
done
grep -v amdgpu.ids $OUT/sweep.txt
