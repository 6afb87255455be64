#!/bin/bash
# Regional heavy-first queues, continued: the producer / consumer one-launch kernel (<= 6000 nodes) with them (RK_HF_REGIONS_PC), sizes 30k-200k;
# fp64 whole trees on k_list_any.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job20
mkdir -p $O
export RK_HF_REGIONS=1
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 0 ]; then unset RK_HF_REGIONS_PC; else export RK_HF_REGIONS_PC=1; fi
    echo -n "pc_regions=$v " | tee -a $O/probe_pc.txt
    timeout 600 python3 tools/pc_ring_probe.py 30000,60000,100000,150000,200000 2>&1 | tail -1 | tee -a $O/probe_pc.txt
  done
done
unset RK_HF_REGIONS_PC
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 0 ]; then unset RK_HF_REGIONS; else export RK_HF_REGIONS=1; fi
    echo -n "regions=$v " | tee -a $O/probe_f64.txt
    timeout 600 python3 tools/pc_ring_probe.py 250000,500000,1000000 float64 2>&1 | tail -1 | tee -a $O/probe_f64.txt
  done
done
