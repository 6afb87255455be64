#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job72; mkdir -p $OUT
for v in trace trace6; do
  export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so
  timeout 300 python3 tools/trace_waves.py $OUT/t_$v.npz 100000 > $OUT/t_$v.log 2>&1; tail -1 $OUT/t_$v.log | cut -c1-150
  timeout 300 python3 tools/trace_digest.py $OUT/t_$v.npz 2>&1 | sed -n 1,4p | cut -c1-330
done
