#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job55; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
for k in 1 2 3 4; do
  RK_PLAN=0 RK_GRAPH_FORKED_MAX=1000000 timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/park_all_$k.log 2>&1; echo "parked, no cap rc=$? $(grep -E 'graph stress ok' $OUT/park_all_$k.log)"
  RK_PLAN=0 timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/park_$k.log 2>&1; echo "parked, cap 64  rc=$? $(grep -E 'graph stress ok' $OUT/park_$k.log)"
done
RK_PLAN_MAX_GROUPS=64 RK_PLAN_REV_MAX_GROUPS=0 RK_GRAPH_FORKED_MAX=1000000 timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/tail.log 2>&1; echo "light-tail plans, parked rc=$? $(grep -E 'graph stress ok' $OUT/tail.log)"
echo "default: $(timeout 600 python3 tools/size_scan.py 2.5e6,4e6 2>&1 | tail -1)"
for i in 1 2 3; do
  timeout 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "native stack\|Error\|error" -B6 -A30 $OUT/run_$i.log | grep -v "^[0-9]*-  File" | head -80; fi
done
