#!/bin/bash
# Onesweep passes in smaller blocks for mid-size sorts (below 2M items), against the library's merge sort: rebuild ms.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job40
mkdir -p $O
make -C examples > /dev/null 2>&1
run() {
  for n in 100000 350000 1000000 2000000; do
    echo -n "$1 $n " | tee -a $O/leapfrog.txt
    timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
  done
}
for rep in 1 2; do
  run default
  for shape in 0 1 2 3; do
    RK_SORT_MIN=0 RK_SORT_SMALL=3000000 RK_SORT_SHAPE=$shape run shape$shape
  done
done
timeout 600 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py -x -q 2>&1 | tail -3
