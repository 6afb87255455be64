#!/bin/bash
# Rebuild after: onesweep launch sequence from 2^20 items, block maxima instead of same-address loads in k_node_counts, child masks by the
# parents (no atomics, no k_popc), G tree levels per launch of the node sums (RK_SUMS_G = 1 / 2 / 3).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job38
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_quadtree.py tests/test_gpu_state_create.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
make -C examples > /dev/null 2>&1
for rep in 1 2; do
for cfg in "g1:RK_SUMS_G=1" "g2:RK_SUMS_G=2" "g3:RK_SUMS_G=3"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for n in 100000 350000 1000000 2000000 4000000; do
    echo -n "$name $n " | tee -a $O/leapfrog.txt
    env $envs timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
  done
done
done
