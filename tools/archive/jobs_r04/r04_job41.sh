#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python3 tools/ordered_probe.py 4000000 2>&1 | grep -v amdgpu
