#!/bin/bash
# Wave timelines of the three regimes of small launches with the round-4 kernels (-DRK_TRACE build).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r04_trace; mkdir -p $OUT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_trace/librakau_amd.so
timeout 300 python3 tools/trace_waves.py $OUT/t100k.npz 100000 > $OUT/t100k.log 2>&1; tail -1 $OUT/t100k.log | cut -c1-200
timeout 300 python3 tools/trace_digest.py $OUT/t100k.npz > $OUT/trace_100k_three_wave_workgroups.txt 2>&1
timeout 300 python3 tools/trace_waves.py $OUT/t1m.npz 1000000 > $OUT/t1m.log 2>&1; tail -1 $OUT/t1m.log | cut -c1-200
timeout 300 python3 tools/trace_digest.py $OUT/t1m.npz > $OUT/trace_1m_one_launch.txt 2>&1
timeout 300 python3 tools/trace_waves.py $OUT/tsh.npz 4000000 0.0 0.125 > $OUT/tsh.log 2>&1; tail -1 $OUT/tsh.log | cut -c1-200
timeout 300 python3 tools/trace_digest.py $OUT/tsh.npz > $OUT/trace_shard0_one_launch.txt 2>&1
rm -f $OUT/*.npz $OUT/*.raw
head -4 $OUT/trace_*.txt | cut -c1-330
