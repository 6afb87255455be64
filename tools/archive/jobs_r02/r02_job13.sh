#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job13
mkdir -p $OUT
cd $ROOT
export PYTHONUNBUFFERED=1
( timeout 900 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_full_size.py tests/test_gpu_leapfrog.py tests/test_gpu_multidevice.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log | cut -c1-300
for c in 1 0 1 0; do RK_SUPER_CACHE=$c timeout 300 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sup cache $c: value', d['value'], 'ms', d['ms_per_step'], 'kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'], 'host', d.get('ms_per_call_host_outputs'))"; done
