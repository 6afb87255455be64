#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job18
mkdir -p $OUT
cd $ROOT
for ch in 16 0 32 8 16; do
  echo "== RK_HOST_CHUNKS=$ch"; RK_HOST_CHUNKS=$ch timeout 300 python3 tools/host_timing.py 2>&1 | grep "^call" | tr '\n' ' '; echo
  RK_HOST_CHUNKS=$ch timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench: value', d['value'], 'kernel_ms', d['kernel_ms'], 'host outputs', d.get('value_host_outputs'), d.get('ms_per_call_host_outputs'))"
done
( timeout 1200 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_reference_tests.py tests/test_gpu_parity_basic.py tests/test_gpu_quadtree.py tests/test_cpp_header.py tests/test_gpu_multidevice.py tests/test_integration_bridge.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
