#!/bin/bash
# Robustness sweep of the final build: the parity / cache / reference-test files under every scheduling knob.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job58
mkdir -p $OUT
cd $ROOT
for cfg in "RK_GRAPH=0" "RK_PLAN=0" "RK_PLAN=2" "RK_PLAN=2 RK_PLAN_MAX_GROUPS=0" "RK_PLAN=2 RK_PLAN_MAX_GROUPS=0 RK_PLAN_REGIONS=0" "RK_SUPER_K=0" "RK_SUPER_K=64" "RK_SUPER_CACHE=0" "RK_SERIAL_CLASSES=1" "RK_BIG_DFS=1" "RK_PC_ALL_BELOW=100000000" "RK_PC_ALL_BELOW=0 RK_PC_R2_BELOW=0" "RK_XCD_MODE=0" "RK_XCD_MODE=2" "RK_HOST_DIRECT=0" "RK_POOL=0"; do
  echo -n "$cfg: " | tee -a $OUT/sweep.txt
  ( env $cfg timeout 900 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_call_caches.py tests/test_gpu_reference_tests.py tests/test_gpu_host_outputs.py tests/test_gpu_device_build.py -m gpu -x -q 2>&1 | tail -1 ) | tee -a $OUT/sweep.txt
done
