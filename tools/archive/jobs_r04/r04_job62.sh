#!/bin/bash
# The pre-pass at small sizes again, now with three-wave workgroups: RK_SUPER_K=0 (members walk from the root) against the default.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for k in 16 0 8 32; do
  echo "RK_SUPER_K=$k $(RK_SUPER_K=$k RK_SUPER_CACHE=0 timeout 300 python3 tools/pc_ring_probe.py 30000,60000,100000,150000,200000 2>&1 | grep -v amdgpu | tail -1)"
done; done
