#!/bin/bash
# Collects the round's judged evidence on the GPU box: kernel-trace stats of the default bench command (overlapped and
# serial-class runs), the default bench line, HBM traffic from PMC passes, leapfrog kernel stats.
# usage (through gpurun): tools/collect_profiles.sh <tag>
set -u
TAG=${1:-v4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlapped -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_overlapped.log 2>&1
RK_SERIAL_CLASSES=1 RK_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leapfrog -- python3 $ROOT/examples/leapfrog.py --nparts 4000000 --steps 20 > $OUT/leapfrog.log 2>&1
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
bash tools/measure_traffic.sh plummer4m_f32 > $OUT/traffic.log 2>&1
find $OUT -name "*kernel_stats.csv" | head; tail -1 $OUT/bench_default.json | cut -c1-400
