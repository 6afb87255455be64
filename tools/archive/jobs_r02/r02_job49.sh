#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job49
mkdir -p $OUT
cd $ROOT
( timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2>/dev/null; cut -c1-330 $OUT/bench.json
