#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job60; mkdir -p $OUT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
timeout 300 python3 tools/trace_waves.py $OUT/t100k.npz 100000 > $OUT/t100k.log 2>&1; tail -1 $OUT/t100k.log
timeout 300 python3 tools/trace_digest.py $OUT/t100k.npz > $OUT/trace_100k_one_launch.txt 2>&1
timeout 300 python3 tools/trace_waves.py $OUT/t1m.npz 1000000 > $OUT/t1m.log 2>&1; tail -1 $OUT/t1m.log
timeout 300 python3 tools/trace_digest.py $OUT/t1m.npz > $OUT/trace_1m_one_launch.txt 2>&1
head -30 $OUT/trace_100k_one_launch.txt
