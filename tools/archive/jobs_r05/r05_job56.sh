#!/bin/bash
# Blocking look-ups of the rebuild: does an active wait in the runtime (ROC_ACTIVE_WAIT_TIMEOUT, us of spinning before the interrupt wait)
# shorten them? examples/leapfrog, rebuild ms.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job56
mkdir -p $O
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 100000 1000000 4000000; do
    for v in 0 50 500; do
      echo -n "wait$v $n " | tee -a $O/leapfrog.txt
      ROC_ACTIVE_WAIT_TIMEOUT=$v timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
