#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job10; mkdir -p $OUT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_stamps/librakau_amd.so
for n in 100000 1000000; do
RK_GRAPH=0 timeout 300 python3 tools/stamps_probe.py $n 4 > $OUT/stamps_$n.txt 2>&1; tail -3 $OUT/stamps_$n.txt
done
