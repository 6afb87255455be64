#!/bin/bash
# Host-output flavours of rk_acc_pot(): pinned output arrays written by the kernels, non-temporal delivery.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job23
mkdir -p $OUT
cd $ROOT
( timeout 1200 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_cpp_header.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log | cut -c1-400
for nt in 1 0; do
  for th in 8 16; do
    echo "== RK_HOST_NT=$nt RK_HOST_THREADS=$th" >> $OUT/bench.txt
    RK_HOST_NT=$nt RK_HOST_THREADS=$th timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d.get(k) for k in ('value', 'kernel_ms', 'ms_per_call_host_outputs', 'ms_per_call_host_outputs_pinned', 'value_host_outputs', 'value_host_outputs_pinned')}, d['host'].get('pinned_equals_pageable'))
" >> $OUT/bench.txt
  done
done
cat $OUT/bench.txt
