#!/bin/bash
# First calls over the launch order made on the device with the tree (k_first_order) against the class lists read backwards.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_call_caches.py tests/test_gpu_leapfrog.py -m gpu -x -q 2>&1 | tail -4
for v in 1 0; do
  echo "RK_FIRST_ORDER=$v $(RK_FIRST_ORDER=$v timeout 300 python3 tools/first_call_probe.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-400)"
done
for v in 1 0; do
  echo "RK_FIRST_ORDER=$v $(RK_FIRST_ORDER=$v timeout 300 python3 tools/first_call_probe.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200)"
done
