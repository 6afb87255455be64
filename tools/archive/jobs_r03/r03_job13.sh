#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job13; mkdir -p $OUT
( time timeout 2400 python3 -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/pytest_gpu.log 2>&1; tail -30 $OUT/pytest_gpu.log
