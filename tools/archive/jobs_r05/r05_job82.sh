#!/bin/bash
# First calls of the forked class sequence captured and replayed through a re-targeted parked executable (RK_GRAPH_FIRST=1) against direct
# launches: the leapfrog harness (every traversal a first call), ms per step / rebuild / traversal.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job82
mkdir -p $O
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 2000000 4000000 8000000; do
    for v in 0 1; do
      echo -n "first$v $n " | tee -a $O/leapfrog.txt
      RK_GRAPH_FIRST=$v timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
timeout 900 python3 -m pytest tests/test_gpu_leapfrog.py tests/test_gpu_call_caches.py -x -q 2>&1 | tail -3
RK_GRAPH_FIRST=1 timeout 900 python3 -m pytest tests/test_gpu_leapfrog.py tests/test_gpu_call_caches.py -x -q 2>&1 | tail -3
