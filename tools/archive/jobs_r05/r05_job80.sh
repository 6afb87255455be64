#!/bin/bash
# The two class kernels that end last (R = 3, R = 1) as two half-lists on two streams each (RK_SPLIT_HEAVY=1): direct launches (first calls,
# the leapfrog harness) and graph replay (bench).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job80
mkdir -p $O
one() {
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pageable-leg > $O/b.json 2> $O/b.err
  python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s device-resident %.3f ms (kernel %.3f) seam %.3f (kernel %.3f)" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"]))
' $O/b.json "$1" || tail -3 $O/b.err
}
make -C examples > /dev/null 2>&1
for rep in 1 2; do
  one "graph"
  RK_SPLIT_HEAVY=1 one "graph, split"
  RK_GRAPH=0 RK_HOST_GRAPH=0 one "direct"
  RK_GRAPH=0 RK_HOST_GRAPH=0 RK_SPLIT_HEAVY=1 one "direct, split"
  for n in 2000000 4000000 8000000; do
    for v in 0 1; do
      echo -n "leapfrog split$v $n "
      RK_SPLIT_HEAVY=$v timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))'
    done
  done
done 2>&1 | tee $O/out.txt
