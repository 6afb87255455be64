#!/bin/bash
# k_first_order for trees of up to 32768 critical nodes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for v in 1 0; do
  echo "RK_FIRST_ORDER=$v $(RK_FIRST_ORDER=$v timeout 300 python3 tools/first_call_probe.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-400)"
done; done
