#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job5; mkdir -p $OUT
bash tools/ab_split.sh 4000000 4 base u2222 u4222 u4422w6 u8422 u4433w 2>&1 | tee $OUT/ab_4m.txt
bash tools/ab_split.sh 1000000 4 base u2222 u4433w 2>&1 | tee $OUT/ab_1m.txt
RK_SERIAL_CLASSES=1 RK_GRAPH=0 bash tools/ab_split.sh 4000000 4 base u2222 u4433w 2>&1 | tee $OUT/ab_4m_serial.txt
