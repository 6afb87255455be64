#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job35
mkdir -p $OUT
cd $ROOT
( timeout 1500 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_call_caches.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | cut -c1-300
for i in 1 2 3; do
timeout 900 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench:', d['value'], d['kernel_ms'], d['ms_per_call_host_outputs'], d['ms_per_call_host_outputs_pinned'])" | tee -a $OUT/bench.txt
done
timeout 300 python3 tools/big_groups_timing.py 500000 4000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench.txt
