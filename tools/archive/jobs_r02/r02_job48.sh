#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job48
mkdir -p $OUT
cd $ROOT
for n in 4000000 100000 500000; do
 for c in 0 1; do
  for t in 1 0; do
    RK_SUPER_CACHE=$c RK_TIMING=$t python3 tools/step_gap.py $n 2>&1 | grep "ms per call" | tee -a $OUT/gap.txt
  done
 done
done
( timeout 900 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_parity_basic.py tests/test_gpu_bench_multirank.py -m gpu -x -q ) 2>&1 | tail -2
