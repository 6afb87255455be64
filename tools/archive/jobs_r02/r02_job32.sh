#!/bin/bash
# Chunked list kernel for critical nodes of more than 256 particles: tests, timing against the scalar walk, the 256M tree
# that runs out of levels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job32
mkdir -p $OUT
cd $ROOT
( timeout 1500 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py tests/test_gpu_device_build.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log | cut -c1-400
for d in 1 0; do
  RK_BIG_DFS=$d timeout 300 python3 tools/big_groups_timing.py 500000 4000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
  RK_BIG_DFS=$d timeout 300 python3 tools/big_groups_timing.py 500000 600 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
done
timeout 900 python3 tools/big_run.py 256e6 2>&1 | grep -v amdgpu.ids | tee $OUT/big_256m.txt
