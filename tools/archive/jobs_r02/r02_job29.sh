#!/bin/bash
# Launch plan sorted by single critical nodes: tests + strong-scaling rehearsal + size sweep.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job29
mkdir -p $OUT
cd $ROOT
( timeout 1500 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_parity_basic.py tests/test_gpu_full_size.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | cut -c1-300
timeout 600 python3 tools/shard_sim.py 4000000 0,0 2>&1 | grep -v amdgpu.ids | tee $OUT/shard_sim.txt
timeout 600 python3 tools/size_sweep.py 3e4,1e5,3.5e5,5e5,1e6,2e6,4e6 2>&1 | grep -v amdgpu.ids | tee $OUT/size_sweep.txt
