#!/bin/bash
# k_list_any as a persistent kernel (RK_ANY_PERSIST=1) against the plain launch: 100k / 350k / 1M and the 8 shards of the 4M tree.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for v in 0 1; do
  echo "RK_ANY_PERSIST=$v $(RK_ANY_PERSIST=$v timeout 600 python3 tools/any_probe.py 2>&1 | grep -v amdgpu | tail -1)"
done; done
for v in 0 1; do
  echo "RK_ANY_PERSIST=$v $(RK_ANY_PERSIST=$v timeout 300 python3 tools/pc_ring_probe.py 250000,500000,750000 2>&1 | grep -v amdgpu | tail -1)"
done
