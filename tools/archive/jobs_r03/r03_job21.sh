#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job21; mkdir -p $OUT
for rep in 1 2; do
for o in default 4312 4321 4132 3412 2431 1234; do
  if [ $o = default ]; then unset RK_CLASS_ORDER; else export RK_CLASS_ORDER=$o; fi
  timeout 300 python3 tools/order_probe.py 2>&1 | tail -1 | tee -a $OUT/order.txt
done; done
