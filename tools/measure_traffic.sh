#!/bin/bash
# HBM traffic of one traversal step from PMC counters (separate --pmc passes, no tracing domains), written to
# profiles/traffic.json for bench.py's roofline.traffic. Run on the GPU box: tools/measure_traffic.sh [workload]
set -u
WL=${1:-plummer4m_f32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-pageable-leg > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-pageable-leg > $OUT/write.log 2>&1
python3 $ROOT/tools/traffic_json.py $OUT $WL $ROOT/gpurun_out/traffic_$WL.json
