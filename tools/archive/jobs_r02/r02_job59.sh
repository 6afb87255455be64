#!/bin/bash
# Exact MAC test from registers (v_readlane) instead of scalar loads of the targets: A/B + parity.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job59
mkdir -p $OUT
cd $ROOT
( timeout 1500 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py tests/test_gpu_call_caches.py -m gpu -x -q ) 2>&1 | tail -2
for rep in 1 2 3; do
  for v in default exp_scalar_exact; do
    if [ $v = default ]; then unset RAKAU_AMD_LIB; else export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
    echo -n "$v: " | tee -a $OUT/ab.txt
    python3 tools/step_gap.py 2>&1 | grep "ms per call" | sed 's/.*back to back/b2b/' | tee -a $OUT/ab.txt
  done
done
for v in default exp_scalar_exact; do
  if [ $v = default ]; then unset RAKAU_AMD_LIB; else export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
  echo "== $v" | tee -a $OUT/ab.txt
  python3 tools/size_sweep.py 3e4,1e5,3.5e5,5e5,1e6,2e6 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
  python3 tools/shard_sim.py 4000000 0,0 2>&1 | grep "N=8" | tee -a $OUT/ab.txt
  python3 tools/size_sweep.py 4e6 float64 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
done
