#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job71; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p100k -- python3 $ROOT/bench.py --workload plummer100k_f32 --no-cpu-baseline > $OUT/b100k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p1m -- python3 $ROOT/bench.py --nparts 1000000 --no-cpu-baseline > $OUT/b1m.log 2>&1
for d in p100k p1m; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv; head -4 $f | cut -c1-220; done
tail -1 $OUT/b1m.log | cut -c1-200
