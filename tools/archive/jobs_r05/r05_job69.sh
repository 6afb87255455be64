#!/bin/bash
# Digit width 9 against 10 bits (same number of passes for 36 and 63 bits: the cost of a pass).
# full sorts (63 bits: 8 / 7 passes). Tests first.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job69
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_leapfrog.py tests/test_gpu_device_build.py tests/test_gpu_quadtree.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
RK_SORT_MIN=0 timeout 1200 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py -x -q 2>&1 | tail -3 | tee -a $O/pytest.txt
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 2000000 4000000 8000000; do
    for v in "p9:RK_SORT_RB=9" "p10:RK_SORT_RB=10" "f9:RK_SORT_RB=9 RK_SORT_PARTIAL=-100" "f10:RK_SORT_RB=10 RK_SORT_PARTIAL=-100"; do
      name=${v%%:*}; envs=${v#*:}
      echo -n "$name $n " | tee -a $O/leapfrog.txt
      env $envs timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
