#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job70; mkdir -p $OUT
for i in 1 2; do
  timeout 900 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/run_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $OUT/run_$i.log | cut -c1-100)"
done
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python3 bench.py 2>/dev/null | cut -c1-200
