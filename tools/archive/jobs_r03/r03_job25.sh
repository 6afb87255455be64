#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job25; mkdir -p $OUT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
timeout 300 python3 tools/trace_waves.py $OUT/trs.npz 4000000 0.0 0.125 > $OUT/trs.log 2>&1
python3 tools/trace_digest.py $OUT/trs.npz > $OUT/trace_any_shard0.txt 2>&1; cat $OUT/trace_any_shard0.txt | head -9; tail -16 $OUT/trace_any_shard0.txt
rm -f $OUT/*.npz
