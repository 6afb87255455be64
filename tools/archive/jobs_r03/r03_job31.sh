#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job31; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_parity_basic.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
for t in 0 0.02 0.05 0.1 0.25 1.0; do
  RK_SPLIT_TOP=$t timeout 300 python3 tools/any_probe.py 2>&1 | tail -1 | sed "s/^/SPLIT_TOP=$t /" | tee -a $OUT/split_top.txt
done; done
