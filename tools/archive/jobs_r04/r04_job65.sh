#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -v amdgpu | tail -6
timeout 900 python3 tools/size_scan.py 1300000,1800000,2300000,2600000 2>&1 | grep -v amdgpu | tail -1
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
