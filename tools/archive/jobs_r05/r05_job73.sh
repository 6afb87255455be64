#!/bin/bash
# examples/leapfrog --reorder K (the caller's arrays moved into tree order every K steps): which K now that the rebuild is cheaper?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
make -C examples > /dev/null 2>&1
for n in 1000000 4000000; do
  for k in 1 2 4 8 16; do
    echo -n "n $n reorder $k: "
    timeout 300 examples/leapfrog --nparts $n --steps 48 --warmup 8 --reorder $k 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))'
  done
done
