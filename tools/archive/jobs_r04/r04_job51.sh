#!/bin/bash
# s_setprio 3 for the first 1/4, 1/8, 1/16 of the heavy-first list in k_pc_any.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
S=30000,60000,100000,150000,200000
for rep in 1 2; do
for v in lib lib_exp_prio2 lib_exp_prio3 lib_exp_prio4; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -1
done; done
