#!/bin/bash
# Launch plan with one spatial region per XCD shared by all class kernels (RK_PLAN_REGIONS=1): timing, parity, L2 traffic.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job46
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for cfg in "RK_PLAN_REGIONS=0" "RK_PLAN_REGIONS=1"; do
    echo -n "$cfg: " | tee -a $OUT/ab.txt
    env $cfg python3 tools/step_gap.py 2>&1 | grep "ms per call" | sed 's/.*back to back/b2b/' | tee -a $OUT/ab.txt
  done
done
( RK_PLAN_REGIONS=1 timeout 900 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_call_caches.py -m gpu -x -q ) 2>&1 | tail -3
RK_PLAN_REGIONS=1 bash tools/measure_traffic.sh plummer4m_f32 > $OUT/traffic.log 2>&1
cp gpurun_out/traffic_plummer4m_f32.json $OUT/traffic_regions.json; head -8 $OUT/traffic_regions.json
