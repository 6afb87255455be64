#!/bin/bash
# Graph replay against direct launches for the one-launch kernels: bench lines (device-resident ms per step) at 100k, 350k, 1M.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for np in 100000 350000 1000000; do
for rep in 1 2; do
for g in 1 0; do
  RK_GRAPH=$g timeout 300 python3 bench.py --workload plummer100k_f32 --nparts $np --no-cpu-baseline --no-pageable-leg 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('n=$np RK_GRAPH=$g seam', d['ms_per_step'], d['kernel_ms'], 'device-resident', d['ms_per_step_device_resident'], d['kernel_ms_device_resident'])"
done; done; done
