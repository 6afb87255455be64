#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for c in 0 32 64 128 256; do
  echo "RK_REV_CHUNK=$c first calls: $(RK_REV_CHUNK=$c timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1 | sed 's/rebuild+traversal [0-9.]* [0-9a-f]*//g')"
  echo "RK_REV_CHUNK=$c repeated: $(RK_REV_CHUNK=$c timeout 300 python3 tools/size_scan.py 1.2e6,1.5e6,2e6 2>&1 | tail -1)"
done
