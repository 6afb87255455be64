#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job47; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
timeout 120 tools/ubench/chain_rate
for k in 1 2; do
  RK_PLAN=0 timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/plain_$k.log 2>&1; echo "plain (no forked capture) rc=$? $(grep -E 'graph stress ok' $OUT/plain_$k.log)"
  timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/default_$k.log 2>&1; echo "default rc=$? $(grep -E 'graph stress ok' $OUT/default_$k.log)"
done
RK_PLAN=0 RK_GRAPH_FORKED=1 timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/forked.log 2>&1; echo "forked capture rc=$? $(grep -E 'iteration|graph stress ok' $OUT/forked.log | tail -1)"
for i in 1 2 3; do
  timeout 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "native stack\|Error\|error" -B6 -A30 $OUT/run_$i.log | grep -v "^[0-9]*-  File" | head -80; fi
done
