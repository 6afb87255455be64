#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job30; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_quadtree.py tests/test_gpu_reference_tests.py tests/test_gpu_call_caches.py tests/test_gpu_config1_100k.py -m gpu -x -q 2>&1 | tail -4
export RK_SUPER_CACHE=0
for rep in 1 2; do
for c in old auto 0 1; do
  unset RK_SUPER_COOP RK_SUPER_BFS
  if [ $c = old ]; then export RK_SUPER_BFS=0; elif [ $c != auto ]; then export RK_SUPER_COOP=$c; fi
  timeout 300 python3 tools/any_probe.py 2>&1 | tail -1 | sed "s/^/PREPASS=$c /" | tee -a $OUT/coop.txt
  timeout 300 python3 tools/run_variant.py 4000000 0 60 2>&1 | tail -1 | sed "s/^/PREPASS=$c /" | tee -a $OUT/coop.txt
done; done
