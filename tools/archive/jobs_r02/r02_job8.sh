#!/bin/bash
# GPU call 8: exact device build, multi-device (aliased) split, clone; P/C auto threshold data; plan at several sizes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job8
mkdir -p $OUT
cd $ROOT
export PYTHONUNBUFFERED=1
( timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_multidevice.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py -m gpu -x -q -s ) > $OUT/pytest.log 2>&1; grep -v amdgpu $OUT/pytest.log | tail -25 | cut -c1-250
RK_PC_MAX_CRIT=0 timeout 600 python3 tools/pc_check.py 30000 200000 350000 500000 > $OUT/pc_check.txt 2>&1; grep -v amdgpu $OUT/pc_check.txt
for n in 100000 500000 1000000; do
 for plan in 0 2; do
  RK_PLAN=$plan timeout 300 python3 bench.py --no-cpu-baseline --nparts $n --steps 30 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n $n plan $plan: value', d['value'], 'kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'])"
 done
done
