#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job8; mkdir -p $OUT
RK_SL_PARTS_BELOW=0 bash tools/ab_split.sh 4000000 4 base exp_minr2 exp_minr2tie exp_w4heavy 2>&1 | tee $OUT/ab_4m_v4.txt
bash tools/ab_split.sh 4000000 2 base exp_minr2 exp_w4heavy 2>&1 | tee $OUT/ab_4m_v2.txt
