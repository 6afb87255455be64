#!/bin/bash
# Stability of graph replay on the host-output path: the multi-device / host-output / call-cache / reference tests five times over.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for i in 1 2 3 4 5; do
  timeout 900 python3 -m pytest tests/test_gpu_multidevice.py tests/test_gpu_host_outputs.py tests/test_gpu_call_caches.py tests/test_gpu_reference_tests.py tests/test_integration_bridge.py tests/test_cpp_header.py -x -q -m gpu 2>&1 | tail -2
done
