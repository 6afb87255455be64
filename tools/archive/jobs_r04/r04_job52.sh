#!/bin/bash
# Two-part pageable call: one pre-pass for both parts against one per part.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2 3; do
for v in 1 2; do
  echo "RK_HOST_SPLIT_PREPASS=$v: $(RK_HOST_SPLIT_PREPASS=$v timeout 300 python3 tools/host_split_probe.py 4000000 2>&1 | grep -v amdgpu | tail -1)"
done; done
for v in 1 2; do
  echo "RK_HOST_SPLIT_PREPASS=$v: $(RK_HOST_SPLIT_PREPASS=$v timeout 300 python3 tools/host_split_probe.py 2000000 2>&1 | grep -v amdgpu | tail -1)"
  echo "RK_HOST_SPLIT_PREPASS=$v: $(RK_HOST_SPLIT_PREPASS=$v timeout 300 python3 tools/host_split_probe.py 4000000 2 2>&1 | grep -v amdgpu | tail -1)"
done
for f in 0.85 0.9; do
  echo "RK_HOST_SPLIT=$f: $(RK_HOST_SPLIT=$f timeout 300 python3 tools/host_split_probe.py 4000000 2>&1 | grep -v amdgpu | tail -1)"
done
timeout 900 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_call_caches.py -m gpu -x -q 2>&1 | tail -3
