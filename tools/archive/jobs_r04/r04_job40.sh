#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 1500 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_reference_tests.py tests/test_gpu_multidevice.py -m gpu -x -q 2>&1 | tail -3
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f (%.4f ms) device %.1f pageable %.1f (%.4f ms) equal %s" % (d["value"], d["ms_per_step"], d["value_device_resident"], d["value_host_outputs_pageable"], d["ms_per_call_host_outputs_pageable"], d["host"]["pinned_equals_pageable"]))'
python3 tools/stress_host_register.py 20 1 2>&1 | tail -2
