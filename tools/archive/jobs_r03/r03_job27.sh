#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job27; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_parity_basic.py tests/test_gpu_bench_multirank.py tests/test_gpu_quadtree.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python3 tools/shard_sim.py 4000000 > $OUT/shard_sim.txt 2>&1; tail -7 $OUT/shard_sim.txt
timeout 300 python3 bench.py --workload plummer100k_f32 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('100k', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
timeout 300 python3 bench.py --nparts 1000000 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('1M', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
