#!/bin/bash
# Problems sized for the HBM of one MI355X: 128M and 256M particles, generated and built on the GPU.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job30
mkdir -p $OUT
cd $ROOT
timeout 600 python3 tools/big_run.py 128e6 2>&1 | grep -v amdgpu.ids | tee $OUT/big_128m.txt
timeout 900 python3 tools/big_run.py 256e6 2>&1 | grep -v amdgpu.ids | tee $OUT/big_256m.txt
