#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job14
mkdir -p $OUT
cd $ROOT
export PYTHONUNBUFFERED=1
( timeout 900 python3 -m pytest tests/test_gpu_device_build.py -m gpu -x -q -s ) > $OUT/pytest.log 2>&1; grep -v amdgpu $OUT/pytest.log | grep -i "4M\|passed\|failed\|Error" | cut -c1-250
