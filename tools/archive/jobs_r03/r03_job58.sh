#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for f in 0 1; do
  echo "RK_FUSE_SUPER=$f first calls: $(RK_FUSE_SUPER=$f timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1)"
  echo "RK_FUSE_SUPER=$f RK_SUPER_CACHE=0 repeated: $(RK_FUSE_SUPER=$f RK_SUPER_CACHE=0 timeout 300 python3 tools/any_probe3.py 2>&1 | tail -1)"
done
