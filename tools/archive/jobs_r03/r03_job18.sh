#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_multidevice.py tests/test_cpp_header.py tests/test_gpu_reference_tests.py -m gpu -x -q --durations=8 2>&1 | tail -14
