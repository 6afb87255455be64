#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job19; mkdir -p $OUT
timeout 300 python3 tools/cold_probe.py > $OUT/cold.txt 2>&1; cat $OUT/cold.txt | tail -12
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
VARIANT=2 RK_GRAPH=0 timeout 300 python3 tools/trace_waves.py $OUT/tr.npz 4000000 > $OUT/tr.log 2>&1
python3 tools/trace_digest.py $OUT/tr.npz > $OUT/trace_v2_4m.txt 2>&1; head -5 $OUT/trace_v2_4m.txt
rm -f $OUT/*.npz
