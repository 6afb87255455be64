#!/bin/bash
# Knobs of the kernel for oversized critical nodes: waves per SIMD it is compiled for, wavefronts per workgroup.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job33
mkdir -p $OUT
cd $ROOT
for v in default bigA bigB bigC; do
  if [ $v = default ]; then unset RAKAU_AMD_LIB; else export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
  echo "== $v" | tee -a $OUT/timing.txt
  timeout 300 python3 tools/big_groups_timing.py 500000 4000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
  timeout 300 python3 tools/big_groups_timing.py 2000000 1000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timing.txt
  timeout 900 python3 tools/big_run.py 256e6 2>&1 | grep -v amdgpu.ids | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('256M unclipped kernel ms', d['kernel_ms'])" | tee -a $OUT/timing.txt
done
