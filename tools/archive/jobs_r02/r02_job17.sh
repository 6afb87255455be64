#!/bin/bash
# Robustness sweep: the parity tests under every scheduling knob of the launch path (results must not depend on any).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job17
mkdir -p $OUT
cd $ROOT
T="tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py tests/test_gpu_call_caches.py tests/test_gpu_leapfrog.py tests/test_golden.py"
i=0
for cfg in "RK_GRAPH=0" "RK_PLAN=2" "RK_PLAN=0 RK_SUPER_CACHE=0" "RK_PC_ALL_BELOW=1000000000" "RK_PC_ALL_BELOW=0 RK_PC_R2_BELOW=0" "RK_SUPER_K=0" "RK_SUPER_K=8 RK_XCD_MODE=0" "RK_XCD_MODE=2 RK_SERIAL_CLASSES=1" "RK_BUILD_EXACT=1" "RAKAU_AMD_CPU_ISA=avx2 RK_HOST_THREADS=1"; do
  i=$((i+1))
  ( env $cfg timeout 600 python3 -m pytest $T -m gpu -x -q ) > $OUT/sweep_$i.log 2>&1
  echo "[$cfg] $(tail -1 $OUT/sweep_$i.log)"
done
( RK_PC_ALL_BELOW=1000000000 timeout 900 python3 -m pytest tests/test_gpu_full_size.py -m gpu -x -q ) > $OUT/sweep_full_pc.log 2>&1; echo "[full size, P/C kernel everywhere] $(tail -1 $OUT/sweep_full_pc.log)"
