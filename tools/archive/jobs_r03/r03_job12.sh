#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job12; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 100000 4000000; do
  RK_SERIAL_CLASSES=1 RK_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -o p -- python3 $ROOT/tools/run_variant.py $n 4 30 > $OUT/run_$n.txt 2>&1
  find $OUT/prof_$n -name "*.csv" ! -name "*kernel_stats*" -delete
done
cd $ROOT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_stamps/librakau_amd.so
for n in 100000 1000000; do
RK_GRAPH=0 timeout 300 python3 tools/stamps_probe.py $n 4 > $OUT/stamps_$n.txt 2>&1; tail -2 $OUT/stamps_$n.txt
done
