#!/bin/bash
# Odd number of sources per split in the tiles of the common list (LDS bank conflicts of the split reads): A/B + parity.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job52
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for v in default evenT; do
    if [ $v = default ]; then unset RAKAU_AMD_LIB; else export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
    echo -n "$v: " | tee -a $OUT/ab.txt
    python3 tools/step_gap.py 2>&1 | grep "ms per call" | sed 's/.*back to back/b2b/' | tee -a $OUT/ab.txt
  done
done
unset RAKAU_AMD_LIB
( timeout 1500 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_call_caches.py tests/test_gpu_full_size.py -m gpu -x -q ) 2>&1 | tail -2
