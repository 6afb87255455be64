#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job29; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_quadtree.py tests/test_gpu_reference_tests.py tests/test_gpu_call_caches.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
for c in 0 1; do
  RK_SUPER_COOP=$c timeout 300 python3 tools/any_probe.py 2>&1 | tail -1 | sed "s/^/COOP=$c /" | tee -a $OUT/coop.txt
  RK_SUPER_COOP=$c timeout 300 python3 tools/run_variant.py 4000000 0 60 2>&1 | tail -1 | sed "s/^/COOP=$c /" | tee -a $OUT/coop.txt
done; done
cd /tmp && export TMPDIR=/tmp
for n in 100000 4000000; do
RK_SERIAL_CLASSES=1 RK_GRAPH=0 RK_SUPER_CACHE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$n -o p -- python3 $ROOT/tools/run_variant.py $n 0 30 > $ROOT/$OUT/run_$n.txt 2>&1
grep -h "k_super" $ROOT/$OUT/prof_$n/*kernel_stats.csv | cut -d, -f1-8 | cut -c1-150
find $ROOT/$OUT/prof_$n -name "*.csv" ! -name "*kernel_stats*" -delete
done
