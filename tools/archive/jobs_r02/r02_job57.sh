#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job57
mkdir -p $OUT
cd $ROOT
( timeout 1500 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_quadtree.py tests/test_gpu_full_size.py -m gpu -x -q ) 2>&1 | tail -2
for n in 1000000 4000000; do ./examples/leapfrog --nparts $n --steps 40 --warmup 10; done 2>&1 | grep -v amdgpu.ids | tee $OUT/leapfrog.txt | cut -c1-330
python3 tools/big_run.py 128e6 50 2>&1 | grep -v amdgpu | cut -c1-200
