#!/bin/bash
# Strong-scaling rehearsal (tools/shard_sim.py) with round 5's kernels; k_list_any compiled for 6 waves per SIMD beside it.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job18
mkdir -p $O
for v in base any6; do
  lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
  echo "== $v"; RAKAU_AMD_LIB=$lib timeout 600 python3 tools/shard_sim.py 2>&1 | grep -v amdgpu | tee $O/shard_sim_$v.txt
done
