#!/bin/bash
# Class kernels of a direct (non-graph) call on ONE stream, all but the first enqueued with hipExtAnyOrderLaunch (no in-order barrier), against
# the fork onto four streams and against graph replay. bench.py 4M (device-resident / seam ms) and the leapfrog harness (every traversal a first call).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job77
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
  one "graph (default)"
  RK_GRAPH=0 RK_HOST_GRAPH=0 one "direct, forked"
  RK_GRAPH=0 RK_HOST_GRAPH=0 RK_SERIAL_CLASSES=1 one "direct, one stream, serial"
  RK_GRAPH=0 RK_HOST_GRAPH=0 RK_SERIAL_CLASSES=1 RK_ANY_ORDER=1 one "direct, one stream, any order"
  for n in 2000000 4000000; do
    for v in "forked:RK_X=0" "anyorder:RK_SERIAL_CLASSES=1 RK_ANY_ORDER=1"; do
      name=${v%%:*}; envs=${v#*:}
      echo -n "leapfrog $name $n "
      env $envs timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))'
    done
  done
done 2>&1 | tee $O/out.txt
