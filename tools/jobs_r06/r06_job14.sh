#!/bin/bash
# (a) k_tail_thr rewritten (loads in flight together, prefix sums): tests + leapfrog; (b) experiment: the build's look-ups of its control
# block polled (hipEventQuery spin) instead of slept on (hipEventSynchronize): lib_exp_spin against the in-tree library.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job14
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_device_build.py -x -q -k first_call 2>&1 | tail -2 | tee $O/tests.txt
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
for rep in 1 2 3; do
  for n in 100000 1000000 4000000; do
    run sleep $n LD_LIBRARY_PATH=$ROOT/rakau_amd/lib:${LD_LIBRARY_PATH:-}
    run spin $n LD_LIBRARY_PATH=$ROOT/rakau_amd/lib_exp_spin:${LD_LIBRARY_PATH:-}
  done
done
