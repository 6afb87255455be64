#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job31
mkdir -p $OUT
cd $ROOT
timeout 900 python3 tools/big_run.py 256e6 50 2>&1 | grep -v amdgpu.ids | tee $OUT/big_256m_clip50.txt
timeout 900 python3 tools/big_run.py 512e6 50 2>&1 | grep -v amdgpu.ids | tee $OUT/big_512m_clip50.txt
