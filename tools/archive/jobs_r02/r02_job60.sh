#!/bin/bash
# Waves per SIMD the class kernels are compiled for, re-checked on the final code.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job60
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  for v in default exp_w3_7 exp_w4_6 exp_w12_6; do
    if [ $v = default ]; then unset RAKAU_AMD_LIB; else export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
    echo -n "$v: " | tee -a $OUT/ab.txt
    python3 tools/step_gap.py 2>&1 | grep "ms per call" | sed 's/.*back to back/b2b/' | tee -a $OUT/ab.txt
  done
done
