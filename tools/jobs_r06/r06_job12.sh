#!/bin/bash
# Class kernels of first-of-their-kind calls through a re-targeted executable graph (launch_classes_retargeted): the leapfrog harness
# (every traversal such a call) against direct forked launches (RK_GRAPH=0 switches every graph off; nothing else in this loop uses one);
# then the whole GPU suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job12
mkdir -p $O
make -C examples > /dev/null 2>&1
run() {
  local label=$1; shift
  local n=$1; shift
  echo -n "$label $n " | tee -a $O/leapfrog.txt
  env "$@" timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f energy %s" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal"), re.search("\"energy[^,]*", l).group(0) if "energy" in l else ""))' | tee -a $O/leapfrog.txt
}
for rep in 1 2 3; do
  for n in 2000000 3000000 4000000 6000000 8000000; do
    run direct $n RK_GRAPH=0
    run retargeted $n RK_GRAPH=1
  done
done
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
