#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job23; mkdir -p $OUT
for rep in 1 2; do
for a in 0 auto 1 2 3; do
  if [ $a = auto ]; then unset RK_ANY; else export RK_ANY=$a; fi
  timeout 300 python3 tools/any_probe.py 2>&1 | tail -1 | tee -a $OUT/any.txt
done; done
