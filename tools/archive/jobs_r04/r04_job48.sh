#!/bin/bash
# Waves per SIMD the producer / consumer kernels are compiled for (RK_PC_W = 4 / 5 / 6) with two and one consumers per workgroup.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
S=30000,100000,150000,200000,250000,350000
for rep in 1 2; do
for v in lib_exp_nc2 lib_exp_nc2w6 lib_exp_nc2w4 lib_exp_nc1w6; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=1 timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -2
done; done
