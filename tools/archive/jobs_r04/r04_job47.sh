#!/bin/bash
# One consumer per workgroup (it takes all R target slots) against two; where k_list_any takes over.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
S=30000,100000,150000,250000,350000,500000,1000000
for rep in 1 2; do
for v in lib_exp_nc2 lib_exp_nc1 lib_exp_nc1nb4 lib_exp_nc2nb4; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=1 timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -2
done; done
RAKAU_AMD_LIB=$ROOT/rakau_amd/lib/librakau_amd.so RK_ANY=3 timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -2
RAKAU_AMD_LIB=$ROOT/rakau_amd/lib/librakau_amd.so timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -2
for v in lib_exp_nc1 lib_exp_nc2; do
RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=1 timeout 300 python3 tools/pc_ring_probe.py 100000,350000 float64 2>&1 | grep -v amdgpu | tail -2
done
RAKAU_AMD_LIB=$ROOT/rakau_amd/lib/librakau_amd.so RK_ANY=3 timeout 300 python3 tools/pc_ring_probe.py 100000,350000 float64 2>&1 | grep -v amdgpu | tail -2
