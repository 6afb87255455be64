#!/bin/bash
# Device code objects compressed in the fat binary (--offload-compress): does the runtime load them, what does it cost at start-up?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in current cmp current cmp; do
  lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
  RAKAU_AMD_LIB=$lib python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "value %.1f device %.1f rk_init_s %s state_create_cold_s %s first_call_ms %s" % (d["value"], d["value_device_resident"], d["host"]["rk_init_s"], d["host"]["state_create_cold_s"], d["host"]["first_call_ms"]))' $v
done
RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_cmp/librakau_amd.so timeout 900 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_quadtree.py tests/test_gpu_device_build.py -m gpu -x -q 2>&1 | tail -3
