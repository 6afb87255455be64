#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE workloads + final suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job55
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in plummer16m_f64 plummer64m_f32 plummer4m_f32_accpot plummer100k_f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -- python3 $ROOT/bench.py --no-cpu-baseline --workload $wl > $OUT/bench_$wl.log 2>&1
  grep "^{" $OUT/bench_$wl.log | cut -c1-220
done
cd $ROOT
( timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
find $OUT -name "*kernel_stats.csv"
