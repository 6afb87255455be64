#!/bin/bash
# k_list_any compiled for 6 waves per SIMD (lib_exp_wany6) against 5 (in-tree), on the regional queues of this round.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job22
mkdir -p $O
for rep in 1 2; do
  for v in base exp_wany6; do
    if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 250000,350000,500000,750000,1000000,1500000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
for v in base exp_wany6; do
  if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
  echo "== $v" | tee -a $O/shards.txt
  RAKAU_AMD_LIB=$lib timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work\|full" | tee -a $O/shards.txt
done
