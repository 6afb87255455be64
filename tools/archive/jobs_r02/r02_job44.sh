#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job44
mkdir -p $OUT
cd $ROOT
for wl in plummer4m_f32 plummer4m_f32_accpot plummer16m_f64 plummer64m_f32 plummer100k_f32; do
    RK_BENCH_DEBUG=1 timeout 900 python3 bench.py --no-cpu-baseline --workload $wl 2> $OUT/err_$wl.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl:', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])" | tee -a $OUT/bench.txt
    grep "pre-loop" $OUT/err_$wl.txt | cut -c1-200 | tee -a $OUT/bench.txt
done
python3 tools/size_sweep.py 2e6 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
RK_PLAN=0 python3 tools/size_sweep.py 2e6 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
