#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job7; mkdir -p $OUT
RK_SL_PARTS_BELOW=0 bash tools/ab_split.sh 4000000 4 base exp_e1 exp_e2 exp_w5 exp_w4 2>&1 | tee $OUT/ab_4m.txt
