#!/bin/bash
# The light-tail arrangement of first calls made on the device (trees of 49 152 .. 250 000 critical nodes): tests, then the leapfrog
# harness (every traversal a first call) with and without it (RK_FIRST_ORDER=0 also switches the small trees' heavy-first order off,
# which does not matter at these sizes).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job4
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_device_build.py -x -q -k "first_call or plummer_tree" 2>&1 | tail -5 | tee $O/tests.txt
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 2000000 3000000 4000000 8000000; do
    for v in 0 1; do
      echo -n "first_order$v $n " | tee -a $O/leapfrog.txt
      RK_FIRST_ORDER=$v timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
timeout 1500 python3 -m pytest tests/test_gpu_leapfrog.py tests/test_gpu_call_caches.py tests/test_gpu_multidevice.py -x -q 2>&1 | tail -3 | tee -a $O/tests.txt
