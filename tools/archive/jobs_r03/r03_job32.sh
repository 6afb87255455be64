#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job32; mkdir -p $OUT
RK_SPLIT_TOP=0.05 RK_ANY=3 timeout 300 python3 tools/run_variant.py 350000 0 12 > $OUT/log.txt 2>&1
grep -v "^  File\|^Extension" $OUT/log.txt | head -30
