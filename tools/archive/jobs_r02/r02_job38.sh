#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job38
mkdir -p $OUT
cd $ROOT
for a in "--steps 20 --warmup 5" "" "--workload plummer100k_f32" "--steps 20 --warmup 5 --reuse-prepass"; do
timeout 900 python3 bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench $a:', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])" | tee -a $OUT/bench.txt
done
( timeout 900 python3 -m pytest tests/test_gpu_bench_multirank.py -m gpu -x -q ) 2>&1 | tail -3
