#!/bin/bash
# Small repeated calls: R = 4 nodes on their own class kernel (6 waves per SIMD) + everything else on a k_list_any compiled for R <= 3
# (80 VGPRs, 6 waves per SIMD) -- RK_ANY=4, an experiment of round 3 (lost then at 5 / 6 waves) -- against the one launch at 5 waves per SIMD.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_rmax3/librakau_amd.so
for rep in 1 2; do
for a in auto 4; do
  if [ $a = auto ]; then unset RK_ANY; else export RK_ANY=$a; fi
  timeout 600 python3 tools/pc_ring_probe.py 350000,500000,750000,1000000,1500000 2>&1 | tail -1
done; done
unset RK_ANY
timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work"
RK_ANY=4 timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work"
