#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job25
mkdir -p $OUT
cd $ROOT
( timeout 1200 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_multidevice.py tests/test_cpp_header.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log | cut -c1-400
for pool in 1 0 1 0; do
echo "== RK_HOST_POOL=$pool" >> $OUT/bench.txt
RK_HOST_POOL=$pool timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d.get(k) for k in ('value', 'kernel_ms', 'ms_per_call_host_outputs', 'ms_per_call_host_outputs_pinned')}, d['host']['kernel_ms_host_outputs'])
" >> $OUT/bench.txt
done
cat $OUT/bench.txt
