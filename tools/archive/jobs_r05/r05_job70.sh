#!/bin/bash
# Rebuilds below 2^20 particles: onesweep passes over the partial key (RK_SORT_MIN=0: 4-5 passes + k_local_sort) against the library's merge sort.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job70
mkdir -p $O
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 200000 350000 600000 1000000; do
    for v in merge partial; do
      if [ $v = partial ]; then export RK_SORT_MIN=0; else unset RK_SORT_MIN; fi
      echo -n "$v $n " | tee -a $O/leapfrog.txt
      timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
