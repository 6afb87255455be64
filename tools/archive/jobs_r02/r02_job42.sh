#!/bin/bash
# Launch plan for every repeated call: tests that drive the caches, all BASELINE workloads A/B against RK_PLAN_TAIL=0.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job42
mkdir -p $OUT
cd $ROOT
( timeout 1500 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_full_size.py tests/test_gpu_parity_basic.py tests/test_gpu_bench_multirank.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | cut -c1-300
for wl in plummer4m_f32 plummer4m_f32_accpot plummer16m_f64 plummer64m_f32; do
  for t in 0.25 0; do
    RK_PLAN_TAIL=$t timeout 900 python3 bench.py --no-cpu-baseline --workload $wl 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl RK_PLAN_TAIL=$t:', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])" | tee -a $OUT/bench.txt
  done
done
python3 tools/size_sweep.py 1e6,2e6 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
