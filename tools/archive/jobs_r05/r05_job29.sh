#!/bin/bash
# Wave timeline of the 4M step replayed from a graph (round 5's kernels, launch order 3-4-1-2): resident waves over time, per-class ends.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job29
mkdir -p $O
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
timeout 300 python3 tools/trace_waves.py $O/tr.npz 4000000 > $O/tr.log 2>&1; tail -1 $O/tr.log
python3 tools/trace_digest.py $O/tr.npz > $O/trace_4m_graph.txt 2>&1; head -24 $O/trace_4m_graph.txt
rm -f $O/tr.npz
