#!/bin/bash
# The native leapfrog harness (device tree rebuild + traversal + integrator per step) on the final build of round 4.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r04_job29
for n in 100000 1000000 4000000; do
  examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a gpurun_out/r04_job29/leapfrog.txt
done
