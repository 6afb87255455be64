#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for a in auto 0 1 3; do
  if [ $a = auto ]; then timeout 300 python3 tools/any_probe3.py 2>&1 | tail -1; else RK_ANY=$a timeout 300 python3 tools/any_probe3.py 2>&1 | tail -1; fi
done
