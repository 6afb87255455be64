#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job28; mkdir -p $OUT
for rep in 1 2; do
for t in 0 1; do
  RK_ANY_TAIL=$t timeout 300 python3 tools/run_variant.py 4000000 0 60 2>&1 | tail -1 | sed "s/^/ANY_TAIL=$t /" | tee -a $OUT/anytail.txt
  RK_ANY_TAIL=$t timeout 300 python3 tools/run_variant.py 2000000 0 60 2>&1 | tail -1 | sed "s/^/ANY_TAIL=$t /" | tee -a $OUT/anytail.txt
done; done
