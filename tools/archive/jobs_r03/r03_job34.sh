#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job34; mkdir -p $OUT
export PYTHONFAULTHANDLER=1
for i in 1 2 3; do
  timeout 900 python3 -m pytest tests -m gpu -q > $OUT/run_$i.log 2>&1; tail -1 $OUT/run_$i.log
  grep -A30 "Fatal Python error" $OUT/run_$i.log | head -50
done
