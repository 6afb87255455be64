#!/bin/bash
# rk_state_create with the conversion on the device: its own tests, then everything that creates states from host trees.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_state_create.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -15
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("bench: value %.1f device %.1f upload_s %s state_create_cold_s %s first_call_ms %s" % (d["value"], d["value_device_resident"], d["host"]["upload_s"], d["host"]["state_create_cold_s"], d["host"]["first_call_ms"]))'
tests/build/cuda_bridge_driver timing 4000000 2>&1 | grep -v amdgpu.ids
