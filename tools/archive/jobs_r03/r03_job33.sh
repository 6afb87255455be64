#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job33; mkdir -p $OUT
export PYTHONFAULTHANDLER=1
for i in 1 2 3 4 5 6; do
  timeout 600 python3 -m pytest tests/test_gpu_call_caches.py -m gpu -x -q > $OUT/run_$i.log 2>&1; tail -1 $OUT/run_$i.log
  grep -B2 -A25 "Fatal Python error\|Segmentation" $OUT/run_$i.log | head -60
done
