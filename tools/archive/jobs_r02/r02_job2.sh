#!/bin/bash
# Round 2, GPU call 2: per-wave timelines of the list kernel (diagnostic -DRK_TRACE build) for the whole 4M problem, for one
# of 8 equal-work shards (core and halo) and for 1M / 100k problems; then the full-size tests again.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job2
mkdir -p $OUT
cd $ROOT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
python3 tools/trace_waves.py $OUT/trace_4m.npz 4000000 > $OUT/trace_4m.log 2>&1
python3 tools/trace_waves.py $OUT/trace_4m_shard0.npz 4000000 0.0 0.125 > $OUT/trace_4m_shard0.log 2>&1
python3 tools/trace_waves.py $OUT/trace_4m_shard3.npz 4000000 0.375 0.5 > $OUT/trace_4m_shard3.log 2>&1
python3 tools/trace_waves.py $OUT/trace_1m.npz 1000000 > $OUT/trace_1m.log 2>&1
python3 tools/trace_waves.py $OUT/trace_100k.npz 100000 > $OUT/trace_100k.log 2>&1
RK_XCD_MODE=0 python3 tools/trace_waves.py $OUT/trace_4m_xcd0.npz 4000000 > $OUT/trace_4m_xcd0.log 2>&1
tail -n 2 $OUT/trace_*.log
unset RAKAU_AMD_LIB
( time timeout 1500 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_bench_multirank.py tests/test_gpu_leapfrog.py -m gpu -x -q --durations=8 ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
