#!/bin/bash
# (a) idle lanes of the dense phase switched off (-DRK_MASK_IDLE=1); (b) packed bodies with ONE consumer wave per producer / consumer
# workgroup (-DRK_PK_BODY=1 -DRK_PC_NCONS=1): kernel ms + result hashes, alternating.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job4
mkdir -p $O
for rep in 1 2; do
  for v in base mask pk_nc1; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 30000,100000,150000,350000,1000000,4000000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
