#!/bin/bash
# (a) light-tail arrangement with sampled thresholds (one block, no global atomics): tests + leapfrog A/B;
# (b) experiment: k_list_any with the records of the next batch requested before the current one is classified (lib_exp_pf).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job8
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
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
  for n in 1970000 3000000 3940000; do
    run first_order0 $n RK_FIRST_ORDER=0
    run first_order1 $n RK_FIRST_ORDER=1
  done
done
for rep in 1 2; do
  for v in base exp_pf; do
    if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 250000,350000,500000,750000,1000000,1500000 2>&1 | tail -1 | tee -a $O/probe_pf.txt
  done
done
for v in base exp_pf; do
  if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
  echo "== $v" | tee -a $O/shard_pf.txt
  RAKAU_AMD_LIB=$lib timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work\|full" | tee -a $O/shard_pf.txt
done
