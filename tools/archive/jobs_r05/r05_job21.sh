#!/bin/bash
# (a) Epilogue without waits between the stores of a lane's targets ("epi") against the tree's build: the seam's call (pinned host
# arrays) beside the device-resident step; (b) the leapfrog harness with the overlapped look-ups of the device build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job21
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-6s value %.1f ms %.4f kernel_ms %s | device-resident %.1f ms %.4f kernel_ms %s | pageable %.1f" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["value_device_resident"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["value_host_outputs_pageable"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2 3; do
  for v in base epi; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "$v" | tee -a $O/bench.txt
  done
done
for n in 100000 350000 1000000 2000000 4000000; do
  timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | cut -c100-330 | tee -a $O/leapfrog.txt
done
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_quadtree.py -x -q 2>&1 | tail -3
