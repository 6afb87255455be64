#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job44; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
run() { # name, env..., then args
  name=$1; shift
  env "$@" timeout 300 python3 tools/stress_graph_capture.py 6000 0 > $OUT/$name.log 2>&1; rc=$?
  echo "$name rc=$rc $(grep -E 'graph stress ok|libamdhip64' $OUT/$name.log | head -2 | tr '\n' ' ')"
}
for k in 1 2 3; do
  run plain_$k RK_PLAN=0
  run default_$k RK_PLAN=1
  run serial_$k RK_PLAN=0 RK_SERIAL_CLASSES=1
done
timeout 300 python3 tools/stress_graph_capture.py 6000 1 > $OUT/stream.log 2>&1; echo "own stream rc=$? $(tail -1 $OUT/stream.log)"
timeout 900 python3 -m pytest tests/test_gpu_device_build.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -12
