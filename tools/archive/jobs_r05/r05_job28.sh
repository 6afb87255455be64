#!/bin/bash
# Order in which the four class kernels are handed to the runtime (they are replayed from one graph): 3124 (default) / 3412 / 4321 / 1234.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job28
mkdir -p $O
for rep in 1 2; do
  for v in base o3412 o3421 o3142 o4312 o3241 o1342; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 2000000,4000000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
