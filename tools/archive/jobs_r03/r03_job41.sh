#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job41; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
# a parent that holds GPU memory and streams like the full suite does, then the subprocess test, many times
for i in $(seq 1 14); do
  timeout 600 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_call_caches.py -m gpu -q -x -p no:cacheprovider -k "light_tail or same_bits or history" > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "native stack\|Segmentation" -B3 -A30 $OUT/run_$i.log | head -90; fi
done
