#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job24; mkdir -p $OUT
for rep in 1 2; do
RK_ANY=0 timeout 300 python3 tools/any_probe2.py 2>&1 | tail -1 | tee -a $OUT/any2.txt
timeout 300 python3 tools/any_probe2.py 2>&1 | tail -1 | tee -a $OUT/any2.txt
RK_PC_ALL_BELOW=8000 timeout 300 python3 tools/any_probe2.py 2>&1 | tail -1 | tee -a $OUT/any2.txt
RK_PC_ALL_BELOW=3500 timeout 300 python3 tools/any_probe2.py 2>&1 | tail -1 | tee -a $OUT/any2.txt
RK_PLAN_MAX_GROUPS=70000 timeout 300 python3 tools/any_probe2.py 2>&1 | tail -1 | tee -a $OUT/any2.txt
RK_PLAN_MAX_GROUPS=70000 RK_ANY=0 timeout 300 python3 tools/any_probe2.py 2>&1 | tail -1 | tee -a $OUT/any2.txt
done
