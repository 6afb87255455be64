#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job56; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
for k in 1 2 3; do
  RK_PLAN=0 timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/park_$k.log 2>&1; echo "parked, cap 64  rc=$? $(grep -E 'graph stress ok' $OUT/park_$k.log)"
done
RK_PLAN_MAX_GROUPS=64 RK_PLAN_REV_MAX_GROUPS=0 timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/tail.log 2>&1; echo "light-tail plans, cap 64 rc=$? $(grep -E 'graph stress ok' $OUT/tail.log)"
timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/default.log 2>&1; echo "default rc=$? $(grep -E 'graph stress ok' $OUT/default.log)"
timeout 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/run_1.log 2>&1; echo "suite rc=$? $(tail -1 $OUT/run_1.log | cut -c1-100)"
