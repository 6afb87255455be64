#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job62; mkdir -p $OUT
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/run.log 2>&1; echo "suite rc=$? $(tail -1 $OUT/run.log | cut -c1-100)"
timeout 600 python3 bench.py > $OUT/bench.json 2>/dev/null; cut -c1-260 $OUT/bench.json
