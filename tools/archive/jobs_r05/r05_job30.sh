#!/bin/bash
# Leftover sources of a tile (fewer than NS) carried to the next tile instead of a masked remainder step per tile (-DRK_CARRY_REMAINDER=1;
# round 2 measured it slower at 7 waves per SIMD): list kernel only, so the bits differ from the producer / consumer kernel's.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job30
mkdir -p $O
for rep in 1 2 3; do
  for v in base carry; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 1000000,4000000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
