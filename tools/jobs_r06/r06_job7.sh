#!/bin/bash
# Size histogram of the light-tail arrangement taken inside k_crit_boxes (one launch less per rebuild): tests, leapfrog A/B.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job7
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_state_create.py tests/test_gpu_multidevice.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
make -C examples > /dev/null 2>&1
run() {
  local label=$1; shift
  local n=$1; shift
  echo -n "$label $n " | tee -a $O/leapfrog.txt
  env "$@" timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
}
for rep in 1 2 3 4; do
  for n in 1970000 3000000 3940000; do
    run first_order0 $n RK_FIRST_ORDER=0
    run first_order1 $n RK_FIRST_ORDER=1
  done
done
