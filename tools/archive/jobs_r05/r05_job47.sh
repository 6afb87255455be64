#!/bin/bash
# k_emit_nodes: one thread per node (RK_EMIT_PER_NODE=1: k_node_starts + k_emit_per_node) against one thread per first particle.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job47
mkdir -p $O
RK_EMIT_PER_NODE=1 timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_quadtree.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 100000 350000 1000000 2000000 4000000; do
    for v in 0 1; do
      echo -n "pernode$v $n " | tee -a $O/leapfrog.txt
      RK_EMIT_PER_NODE=$v timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done
