#!/bin/bash
# Ring hand-off between producer and consumers (RK_PC_NB = 4 / 8 tile buffers) against the barrier hand-off (2).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for v in lib lib_exp_nb4 lib_exp_nb8; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=1 timeout 300 python3 tools/pc_ring_probe.py 2>&1 | grep -v amdgpu | tail -2
done; done
for v in lib lib_exp_nb4 lib_exp_nb8; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so timeout 300 python3 tools/pc_ring_probe.py 100000,500000 2>&1 | grep -v amdgpu | tail -2
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY=1 timeout 300 python3 tools/pc_ring_probe.py 100000 float64 2>&1 | grep -v amdgpu | tail -2
done
