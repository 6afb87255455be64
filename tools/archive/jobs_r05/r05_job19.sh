#!/bin/bash
# Occupancy / unrolling retune on top of round 5's kernels: R = 4 at 6 waves per SIMD, R = 1 unrolled 2 instead of 4, R = 2 unrolled 1
# instead of 2. Four alternating rounds of the device-resident 4M step (tools/pc_ring_probe.py: kernel ms + hash).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job19
mkdir -p $O
for rep in 1 2 3 4; do
  for v in base r4w6 unr12 unr21; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 1000000,4000000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
