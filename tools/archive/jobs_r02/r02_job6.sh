#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job6
mkdir -p $OUT
cd $ROOT
for ch in 1 4; do RK_HOST_TIMING=1 RK_HOST_CHUNKS=$ch timeout 300 python3 tools/host_timing.py > $OUT/host_timing_$ch.log 2>&1; echo "== chunks $ch"; grep -v amdgpu $OUT/host_timing_$ch.log | tail -12 | cut -c1-420; done
