#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job45; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1 RK_PLAN=0
run() { # name, stream flag, env...
  name=$1; sflag=$2; shift; shift
  env "$@" timeout 300 python3 tools/stress_graph_capture.py 6000 $sflag > $OUT/$name.log 2>&1; rc=$?
  echo "$name rc=$rc last: $(grep -E 'iteration|graph stress ok|Error' $OUT/$name.log | tail -1)"
}
run base 0 A=1
run base_again 0 A=1
run keep_execs 0 RK_GRAPH_DBG=1
run sync_before_capture 0 RK_GRAPH_DBG=2
run sync_before_launch 0 RK_GRAPH_DBG=4
run sync_before_destroy 0 RK_GRAPH_DBG=16
run own_stream 1 A=1
run no_timing_events 0 STRESS_TIMING=0
