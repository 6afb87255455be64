#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for t in 8 12; do
echo "RK_HOST_THREADS=$t"
RK_HOST_THREADS=$t python3 tools/host_split_probe.py 4000000 0 2>&1 | tail -1
RK_HOST_THREADS=$t python3 tools/host_split_probe.py 2000000 0 2>&1 | tail -1
RK_HOST_THREADS=$t python3 tools/host_split_probe.py 4000000 2 2>&1 | tail -1
done
