#!/bin/bash
# Pageable output arrays (two-part staged delivery) with and without graph replay of the two parts.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2 3; do
for v in 0 1; do
echo "RK_HOST_GRAPH=$v: $(RK_HOST_GRAPH=$v python3 tools/host_split_probe.py 4000000 0 2>&1 | tail -1 | cut -c24-)"
done; done
for v in 0 1; do
echo "RK_HOST_GRAPH=$v: $(RK_HOST_GRAPH=$v python3 tools/host_split_probe.py 2000000 0 2>&1 | tail -1 | cut -c24-)"
echo "RK_HOST_GRAPH=$v: $(RK_HOST_GRAPH=$v python3 tools/host_split_probe.py 4000000 2 2>&1 | tail -1 | cut -c24-)"
done
timeout 600 python3 -m pytest tests/test_gpu_host_outputs.py -x -q 2>&1 | tail -2
