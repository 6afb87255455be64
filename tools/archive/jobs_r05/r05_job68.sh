#!/bin/bash
# Digit width of the onesweep passes: 8 bits (the library's) against 9 (fewer passes): partial-key rebuilds (36 bits: 5 / 4 passes) and
# full sorts (63 bits: 8 / 7 passes). Tests first.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job68
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_leapfrog.py tests/test_gpu_device_build.py tests/test_gpu_quadtree.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
RK_SORT_MIN=0 timeout 1200 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py -x -q 2>&1 | tail -3 | tee -a $O/pytest.txt
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 2000000 4000000 8000000; do
    for v in "p8:RK_SORT_RB=8" "p9:RK_SORT_RB=9" "f8:RK_SORT_RB=8 RK_SORT_PARTIAL=-100" "f9:RK_SORT_RB=9 RK_SORT_PARTIAL=-100"; do
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
