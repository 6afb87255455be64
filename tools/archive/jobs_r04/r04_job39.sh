#!/bin/bash
# Pageable host outputs in two parts (the first delivered while the second is traversed): fraction sweep at 4M, other sizes, tests.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
  for f in 0 0.7 0.8 0.85 0.9; do RK_HOST_SPLIT=$f timeout 300 python3 tools/host_split_probe.py 4000000 2>&1 | grep RK_HOST_SPLIT; done
done
for f in 0 0.8; do RK_HOST_SPLIT=$f timeout 300 python3 tools/host_split_probe.py 2000000 2>&1 | grep RK_HOST_SPLIT; RK_HOST_SPLIT=$f timeout 300 python3 tools/host_split_probe.py 4000000 2 2>&1 | grep RK_HOST_SPLIT; done
timeout 900 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_multidevice.py tests/test_gpu_full_size.py tests/test_integration_bridge.py tests/test_cpp_header.py -m gpu -x -q 2>&1 | tail -3
