#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_device_build.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15
