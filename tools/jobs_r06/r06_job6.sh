#!/bin/bash
# (a) tests + leapfrog of the final form of the device-made light-tail arrangement (thresholds inside k_bin_count; precedence over the
#     one-launch first call from 49 152 nodes); (b) wave timelines on this round's kernels (-DRK_TRACE build): shard 0 of 8 of the 4M
#     tree, the 1M tree, 100k.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job6
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
  for n in 1970000 3940000; do
    run first_order0 $n RK_FIRST_ORDER=0
    run first_order1 $n RK_FIRST_ORDER=1
  done
done
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
timeout 600 python3 tools/trace_waves.py $O/trace_shard0.npz 4000000 0.0 0.125 2>&1 | tail -1
timeout 300 python3 tools/trace_digest.py $O/trace_shard0.npz > $O/trace_shard0_one_launch.txt 2>&1
timeout 600 python3 tools/trace_waves.py $O/trace_1m.npz 1000000 2>&1 | tail -1
timeout 300 python3 tools/trace_digest.py $O/trace_1m.npz > $O/trace_1m_one_launch.txt 2>&1
rm -f $O/*.npz $O/*.raw
