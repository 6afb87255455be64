#!/bin/bash
# After raising the one-launch limits (plans to 45000 critical nodes, first-call order made on the device to 49152): repeated calls,
# first calls in the leapfrog harness, the device-build tests.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job33
mkdir -p $O
timeout 600 python3 tools/pc_ring_probe.py 1000000,1250000,1500000,1750000,2000000 2>&1 | tail -1 | tee $O/probe.txt
for n in 1000000 1300000 1600000 2000000; do
  for fo in 1 0; do echo "RK_FIRST_ORDER=$fo $(RK_FIRST_ORDER=$fo timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | cut -c170-330)" | tee -a $O/leapfrog.txt; done
done
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_call_caches.py tests/test_gpu_leapfrog.py -x -q 2>&1 | tail -3
