#!/bin/bash
# Cost of a call through the stateless CUDA seam with and without the life-time hooks.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r04_job22
for n in 1000000 4000000; do tests/build/cuda_bridge_driver timing $n 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_job22/timing.txt; done
