#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job24
mkdir -p $OUT
cd $ROOT
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d.get(k) for k in ('value', 'ms_per_step', 'kernel_ms', 'ms_per_call_host_outputs', 'ms_per_call_host_outputs_pinned')}, d['host'])
" >> $OUT/bench.txt
done
cat $OUT/bench.txt
