#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job20; mkdir -p $OUT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
VARIANT=2 RK_GRAPH=0 timeout 300 python3 tools/trace_waves.py $OUT/tr.npz 4000000 > $OUT/tr.log 2>&1
python3 tools/trace_digest.py $OUT/tr.npz > $OUT/trace_v2_4m.txt 2>&1; head -6 $OUT/trace_v2_4m.txt
timeout 300 python3 tools/trace_waves.py $OUT/trs.npz 4000000 0.0 0.125 > $OUT/trs.log 2>&1
python3 tools/trace_digest.py $OUT/trs.npz > $OUT/trace_v0_shard0.txt 2>&1; cat $OUT/trace_v0_shard0.txt | head -30; tail -5 $OUT/trace_v0_shard0.txt
VARIANT=2 timeout 300 python3 tools/trace_waves.py $OUT/trs2.npz 4000000 0.0 0.125 > $OUT/trs2.log 2>&1
python3 tools/trace_digest.py $OUT/trs2.npz > $OUT/trace_v2_shard0.txt 2>&1; head -6 $OUT/trace_v2_shard0.txt; tail -5 $OUT/trace_v2_shard0.txt
rm -f $OUT/*.npz
