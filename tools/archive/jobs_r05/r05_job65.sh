#!/bin/bash
# Partial-key sort in rebuilds (code bits of the previous tree's depth + 1, the leaves' insides ordered by k_local_sort): tests, then
# rebuild ms against the full sort (RK_SORT_PARTIAL=-100), same library, interleaved.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job65
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_leapfrog.py tests/test_gpu_device_build.py tests/test_gpu_quadtree.py -x -q 2>&1 | tail -15 | tee $O/pytest.txt
make -C examples > /dev/null 2>&1
RK_SORT_TRACE=1 timeout 300 examples/leapfrog --nparts 4000000 --steps 6 --warmup 2 2>&1 | grep rk_build | tail -4
for rep in 1 2 3; do
  for n in 1000000 2000000 4000000 8000000; do
    for v in full partial; do
      if [ $v = full ]; then export RK_SORT_PARTIAL=-100; else unset RK_SORT_PARTIAL; fi
      echo -n "$v $n " | tee -a $O/leapfrog.txt
      timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
