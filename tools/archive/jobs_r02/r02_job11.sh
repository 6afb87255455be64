#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job11
mkdir -p $OUT
cd $ROOT
export PYTHONUNBUFFERED=1
echo "== W=5"; timeout 600 python3 tools/pc_check.py 30000 100000 350000 1000000 4000000 2>&1 | grep -v amdgpu
for v in pcw6 pcw4; do echo "== $v"; RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so timeout 600 python3 tools/pc_check.py 100000 1000000 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/shard_sim.py 4000000 2,3 2>&1 | grep -v amdgpu
( timeout 900 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py tests/test_golden.py tests/test_gpu_leapfrog.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
