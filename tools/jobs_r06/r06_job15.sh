#!/bin/bash
# k_tail_thr rewritten + the harness's free-running step time: tests of the device build / leapfrog, then the harness at seven sizes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job15
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py -x -q 2>&1 | tail -2 | tee $O/tests.txt
make -C examples > /dev/null 2>&1
for n in 100000 350000 1000000 2000000 3000000 4000000 8000000; do
  timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("nparts %d step %.4f rebuild %.4f traversal %.4f free-running step %.4f" % (g("nparts"), g("ms_per_step"), g("ms_rebuild"), g("ms_traversal"), g("ms_per_step_free_running")))' | tee -a $O/leapfrog.txt
done
