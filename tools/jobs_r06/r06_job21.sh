#!/bin/bash
# First-call launch order of small trees as eight per-XCD regional queues (in-tree) against one list sorted by size (lib_exp_prev):
# tests, then the leapfrog harness (every traversal a first call) at 100k-1.5M particles.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job21
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_call_caches.py tests/test_gpu_multidevice.py -x -q 2>&1 | tail -2 | tee $O/tests.txt
make -C examples > /dev/null 2>&1
run() {
  local label=$1; shift
  local n=$1; shift
  echo -n "$label $n " | tee -a $O/leapfrog.txt
  env "$@" timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f free-running %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal"), g("ms_per_step_free_running")))' | tee -a $O/leapfrog.txt
}
for rep in 1 2 3; do
  for n in 100000 200000 350000 700000 1000000 1500000; do
    run one_list $n LD_LIBRARY_PATH=$ROOT/rakau_amd/lib_exp_prev:${LD_LIBRARY_PATH:-}
    run queues $n LD_LIBRARY_PATH=$ROOT/rakau_amd/lib:${LD_LIBRARY_PATH:-}
  done
done
