#!/bin/bash
# Experiment: the heavy-first list of k_list_any as eight per-XCD regional queues (RK_HF_REGIONS=1: each XCD's L2 sees one eighth of
# the range) against chunks of 16 list entries dealt round-robin (default).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job19
mkdir -p $O
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 0 ]; then unset RK_HF_REGIONS; else export RK_HF_REGIONS=1; fi
    echo -n "regions=$v " | tee -a $O/probe.txt
    timeout 600 python3 tools/pc_ring_probe.py 250000,350000,500000,750000,1000000,1500000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
for v in 0 1; do
  if [ $v = 0 ]; then unset RK_HF_REGIONS; else export RK_HF_REGIONS=1; fi
  echo "== regions=$v" | tee -a $O/shards.txt
  timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work\|full" | tee -a $O/shards.txt
done
