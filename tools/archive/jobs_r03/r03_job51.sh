#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job51; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
RK_ANY_FIRST_MAX=60000 timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1
for n in 100000 1000000 4000000; do
  for f in 0 1; do
    echo "leapfrog n=$n RK_ANY_FIRST=$f: $(RK_ANY_FIRST=$f timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | cut -c1-260)"
  done
done
for i in 1 2; do
  timeout 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "native stack\|Error\|error" -B6 -A30 $OUT/run_$i.log | grep -v "^[0-9]*-  File" | head -80; fi
done
