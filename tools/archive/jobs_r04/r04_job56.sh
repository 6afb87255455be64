#!/bin/bash
# What the heavy-first order and the graph replay are each worth at 100k-200k with the three-wave workgroups.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
S=100000,150000,200000
echo "plan + graph:      $(timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -1)"
echo "no plan, graph:    $(RK_PLAN=0 timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -1)"
echo "plan, no graph:    $(RK_GRAPH=0 timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -1)"
echo "no plan, no graph: $(RK_PLAN=0 RK_GRAPH=0 timeout 300 python3 tools/pc_ring_probe.py $S 2>&1 | grep -v amdgpu | tail -1)"
