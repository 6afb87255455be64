#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job43
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for wl in plummer4m_f32 plummer16m_f64 plummer64m_f32; do
  for pl in 1 0; do
    RK_PLAN=$pl timeout 900 python3 bench.py --no-cpu-baseline --workload $wl 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl RK_PLAN=$pl:', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])" | tee -a $OUT/bench.txt
  done
done
done
