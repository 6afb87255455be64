#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job63; mkdir -p $OUT
python3 -c "import torch; print(torch.cuda.is_available())"
timeout 1200 python3 -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $OUT/run.log 2>&1; echo "suite rc=$? $(tail -1 $OUT/run.log | cut -c1-100)"
grep -A14 "slowest" $OUT/run.log
