#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job37; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
for i in 1 2 3 4 5 6; do
  timeout 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "native stack" -B6 -A40 $OUT/run_$i.log | grep -v "^[0-9]*-  File" | head -80; fi
done
