#!/bin/bash
# Rebuild after: k_node_counts with one LDS atomic per wavefront, critical-node boxes by one wavefront per node, k_box folded into k_encode.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job14
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_state_create.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for n in 100000 350000 1000000 4000000; do
  timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | cut -c100-330 | tee -a $O/leapfrog.txt
done
