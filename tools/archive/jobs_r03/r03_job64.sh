#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job64; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
for i in 1 2 3 4 5; do
  timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "native stack\|Error\|error\|FAILED" -B6 -A30 $OUT/run_$i.log | grep -v "^[0-9]*-  File" | head -80; fi
done
MALLOC_MMAP_THRESHOLD_=33554432 timeout 200 python3 tools/stress_host_register.py 40 21 2>&1 | tail -1
