#!/bin/bash
# Fewer consumer waves per workgroup (RK_PC_NCONS = 2 / 3: consumers take several target slots) against 4, with and without the ring.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for v in lib lib_exp_nc2 lib_exp_nc3 lib_exp_nc2nb4; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=1 timeout 300 python3 tools/pc_ring_probe.py 30000,100000,150000,250000,350000 2>&1 | grep -v amdgpu | tail -2
done; done
for v in lib lib_exp_nc2; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=1 timeout 300 python3 tools/pc_ring_probe.py 100000 float64 2>&1 | grep -v amdgpu | tail -2
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=0 timeout 300 python3 tools/pc_ring_probe.py 100000 2>&1 | grep -v amdgpu | tail -2
done
