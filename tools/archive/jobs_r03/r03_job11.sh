#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job11; mkdir -p $OUT
SIZES=1e5,1e6,4e6 timeout 600 python3 tools/split_check.py > $OUT/split_check.txt 2>&1
grep -E "FAIL|CHECK|kernel ms|Error|error" $OUT/split_check.txt | tail -30
cd /tmp && export TMPDIR=/tmp
for n in 100000 4000000; do
  RK_SERIAL_CLASSES=1 RK_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -o p -- python3 $ROOT/tools/run_variant.py $n 4 30 > $OUT/run_$n.txt 2>&1
  find $OUT/prof_$n -name "*.csv" ! -name "*kernel_stats*" -delete
done
