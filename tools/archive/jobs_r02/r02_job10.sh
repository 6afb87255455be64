#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job10
mkdir -p $OUT
cd $ROOT
export PYTHONUNBUFFERED=1
echo "== prefetching producer (KC=2)"; RK_PC_MAX_CRIT=0 timeout 600 python3 tools/pc_check.py 30000 100000 350000 1000000 2>&1 | grep -v amdgpu
echo "== no prefetch"; RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_nopf/librakau_amd.so RK_PC_MAX_CRIT=0 timeout 600 python3 tools/pc_check.py 30000 100000 350000 1000000 2>&1 | grep -v amdgpu
( timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py tests/test_golden.py -m gpu -x -q -s ) > $OUT/pytest.log 2>&1; grep -v amdgpu $OUT/pytest.log | grep -i "4M\|passed\|failed\|Error" | cut -c1-250
timeout 600 python3 tools/shard_sim.py 4000000 0,4 2>&1 | grep -v amdgpu
