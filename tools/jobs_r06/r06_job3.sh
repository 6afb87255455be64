#!/bin/bash
# What a launch plan made on the device could give FIRST calls beyond 49 152 critical nodes: repeated calls without graph replay,
# with (RK_PLAN=1: light-tail plan, one spatial region per XCD) and without (RK_PLAN=0: class lists in Morton order, one slice per XCD)
# the host-made plan. Kernel ms by events.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job3
mkdir -p $O
for rep in 1 2; do
  for p in 0 1; do
    echo -n "RK_GRAPH=0 RK_PLAN=$p: " | tee -a $O/plan_gain.txt
    RK_GRAPH=0 RK_PLAN=$p timeout 900 python3 tools/pc_ring_probe.py 2000000,3000000,4000000 2>&1 | tail -1 | tee -a $O/plan_gain.txt
  done
done
echo -n "RK_GRAPH=1 RK_PLAN=1: " | tee -a $O/plan_gain.txt
timeout 900 python3 tools/pc_ring_probe.py 2000000,3000000,4000000 2>&1 | tail -1 | tee -a $O/plan_gain.txt
