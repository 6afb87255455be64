#!/bin/bash
# Rebuild: look-ups of the control block as asynchronous copies + events, with the pyramid passes / the centres of mass enqueued behind them (now) against blocking copies (prev).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job57
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_quadtree.py tests/test_gpu_state_create.py tests/test_gpu_call_caches.py tests/test_gpu_multidevice.py tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 100000 350000 1000000 2000000 4000000; do
    for v in prev now; do
      if [ $v = now ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$ROOT/rakau_amd/lib_exp_$v; fi
      echo -n "$v $n " | tee -a $O/leapfrog.txt
      timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
