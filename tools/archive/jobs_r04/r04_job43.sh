#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python3 tools/ordered_probe.py 4000000 2>&1 | grep -v amdgpu
timeout 600 python3 tools/ordered_probe.py 100000 2>&1 | grep -v amdgpu
timeout 1500 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_cpp_header.py tests/test_gpu_reference_tests.py tests/test_gpu_multidevice.py tests/test_gpu_quadtree.py tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py -m gpu -x -q 2>&1 | tail -8
