#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job47
mkdir -p $OUT
cd $ROOT
( timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | cut -c1-300
timeout 900 python3 bench.py > $OUT/bench.json 2>/dev/null; cut -c1-400 $OUT/bench.json
python3 tools/size_sweep.py 1e6,2e6,8e6 2>&1 | grep -v amdgpu
