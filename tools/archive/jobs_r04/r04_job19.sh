#!/bin/bash
# The whole GPU suite five times in a row (executable-graph cache, cross-check library loaded on demand, CUDA-seam driver ...).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r04_job19
for i in 1 2 3 4 5; do
  timeout 1500 python3 -m pytest tests -m gpu -q -x > gpurun_out/r04_job19/run$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/r04_job19/run$i.log)"
done
