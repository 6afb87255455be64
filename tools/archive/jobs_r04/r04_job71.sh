#!/bin/bash
# Heavy-first plans sorted by the interaction census of the nodes instead of their sizes (RK_PLAN_CENSUS=1, experimental build).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_census/librakau_amd.so
for rep in 1 2; do
for v in 0 1; do
  echo "RK_PLAN_CENSUS=$v $(RK_PLAN_CENSUS=$v timeout 600 python3 tools/any_probe.py 2>&1 | grep -v amdgpu | tail -1)"
done; done
for v in 0 1; do
  echo "RK_PLAN_CENSUS=$v $(RK_PLAN_CENSUS=$v timeout 300 python3 tools/pc_ring_probe.py 30000,150000,250000,500000,750000 2>&1 | grep -v amdgpu | tail -1)"
done
