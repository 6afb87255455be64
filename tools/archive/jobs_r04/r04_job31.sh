#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_xcheck_library.py -m gpu -x -q 2>&1 | tail -12
