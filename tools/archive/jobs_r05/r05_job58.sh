#!/bin/bash
# The library's merge sort below 2^20 items with tiles of 1024 (its own choice) / 2048 / 4096 items: fewer merge passes, no copy launches
# when their number is even. Rebuild ms, same library, interleaved.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job58
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 30000 100000 200000 350000 600000 1000000; do
    for v in 1024 2048 4096 0; do
      echo -n "tile$v $n " | tee -a $O/leapfrog.txt
      RK_SORT_TILE=$v timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
