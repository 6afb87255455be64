#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_multidevice.py -m gpu -x -q 2>&1 | tail -40
RK_HOST_GRAPH=0 timeout 900 python3 -m pytest tests/test_gpu_multidevice.py -m gpu -x -q 2>&1 | tail -5
