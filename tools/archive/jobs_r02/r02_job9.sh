#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job9
mkdir -p $OUT
cd $ROOT
export PYTHONUNBUFFERED=1
( timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_multidevice.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py tests/test_gpu_leapfrog.py tests/test_cpp_header.py -m gpu -x -q -s ) > $OUT/pytest.log 2>&1; grep -v amdgpu $OUT/pytest.log | grep -i "4M\|passed\|failed\|Error" | cut -c1-250
for n in 2000000 4000000; do
 for plan in 0 2; do
  for rep in 1 2; do
  RK_PLAN=$plan timeout 300 python3 bench.py --no-cpu-baseline --nparts $n --steps 30 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n $n plan $plan: value', d['value'], 'kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'])"
  done
 done
done
