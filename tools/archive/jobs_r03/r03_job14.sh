#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job14; mkdir -p $OUT
python3 tools/cold_probe.py > $OUT/cold1.txt 2>&1; cat $OUT/cold1.txt
python3 tools/cold_probe.py > $OUT/cold2.txt 2>&1; grep -E "device_count|first rk_state|first rk_acc|rk_state_create \(4|acc_pot #1" $OUT/cold2.txt
HIP_ENABLE_DEFERRED_LOADING=1 python3 tools/cold_probe.py > $OUT/cold3.txt 2>&1; grep -E "device_count|first rk_state|first rk_acc|rk_state_create \(4|acc_pot #1" $OUT/cold3.txt
