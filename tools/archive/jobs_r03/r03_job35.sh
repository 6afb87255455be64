#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job35; mkdir -p $OUT
export PYTHONFAULTHANDLER=1
for i in $(seq 1 25); do
  timeout 300 python3 -m pytest tests/test_gpu_quadtree.py tests/test_gpu_reference_tests.py -m gpu -x -q -s > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-80)"
  if [ $rc -ne 0 ]; then grep -v "^  File" $OUT/run_$i.log | tail -25; fi
done
