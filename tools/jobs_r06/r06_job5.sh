#!/bin/bash
# Light-tail arrangement of first calls, second build (thresholds from a one-block kernel): leapfrog with / without it; at 1.7M-2.2M
# particles (46k-60k nodes) also against the one launch over the class lists read backwards (RK_ANY_FIRST=0 leaves the tail arrangement).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job5
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_device_build.py -x -q -k "first_call" 2>&1 | tail -3 | tee $O/tests.txt
make -C examples > /dev/null 2>&1
run() { # label, env..., n
  local label=$1; shift
  local n=$1; shift
  echo -n "$label $n " | tee -a $O/leapfrog.txt
  env "$@" timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
}
for rep in 1 2 3; do
  for n in 3000000 4000000 6000000; do
    run first_order0 $n RK_FIRST_ORDER=0
    run first_order1 $n RK_FIRST_ORDER=1
  done
  for n in 1900000 2200000; do
    run any_first1 $n RK_ANY_FIRST=1
    run any_first0_tail $n RK_ANY_FIRST=0
    run any_first0_notail $n RK_ANY_FIRST=0 RK_FIRST_ORDER=0
  done
done
