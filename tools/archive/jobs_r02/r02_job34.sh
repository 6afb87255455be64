#!/bin/bash
# Whole GPU suite + default bench after the chunked kernel for oversized nodes, the per-node launch plan and the pinned outputs.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job34
mkdir -p $OUT
cd $ROOT
( time timeout 2400 python3 -m pytest tests -m gpu -x -q --durations=10 ) > $OUT/pytest.log 2>&1; tail -20 $OUT/pytest.log | cut -c1-300
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-1500
timeout 900 python3 bench.py --reuse-prepass --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('reuse-prepass:', d['value'], d['kernel_ms'])"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
