#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job37
mkdir -p $OUT
cd $ROOT
for c in 1 0; do for g in 1 0; do RK_SUPER_CACHE=$c RK_GRAPH=$g python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/gap.txt; done; done
RK_SUPER_CACHE=0 RK_EVENTS=0 python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/gap.txt
RK_SUPER_CACHE=0 RK_SERIAL_CLASSES=1 python3 tools/step_gap.py 2>&1 | grep "ms per call" | tee -a $OUT/gap.txt
