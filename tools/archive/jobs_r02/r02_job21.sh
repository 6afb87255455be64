#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job21
mkdir -p $OUT
cd $ROOT
( timeout 900 python3 -m pytest tests/test_gpu_bench_multirank.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log | cut -c1-300
