#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job56
mkdir -p $OUT
cd $ROOT
( timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
timeout 900 python3 bench.py --workload plummer16m_f64 --no-cpu-baseline 2>/dev/null | cut -c1-200
