#!/bin/bash
# Round 3, GPU call 2: per-kernel times of the split traversal (k_lists / k_dense) at 100k and 4M.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 100000 4000000; do
  RK_SERIAL_CLASSES=1 RK_GRAPH=0 rocprofv3 --kernel-trace --stats -d $OUT/prof_$n -o p -- python3 $ROOT/tools/run_variant.py $n 4 30 > $OUT/run_$n.txt 2>&1
  f=$(find $OUT/prof_$n -name "*kernel_stats.csv" | head -1)
  echo "== n=$n"; tail -1 $OUT/run_$n.txt; head -12 $f | cut -c1-200
done
