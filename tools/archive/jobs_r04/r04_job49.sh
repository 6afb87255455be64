#!/bin/bash
# Three-wave producer / consumer workgroups as the default: thresholds (fp32 / fp64), the benches of the small sizes, the suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for a in auto 1 3; do
  if [ $a = auto ]; then unset RK_ANY; else export RK_ANY=$a; fi
  timeout 300 python3 tools/pc_ring_probe.py 100000,150000,200000,250000,300000 2>&1 | grep -v amdgpu | tail -1
  timeout 300 python3 tools/pc_ring_probe.py 60000,100000,150000,200000,250000 float64 2>&1 | grep -v amdgpu | tail -1
done
unset RK_ANY
timeout 300 python3 tools/first_call_probe.py 2>&1 | grep -v amdgpu | tail -8
timeout 300 python3 bench.py --workload plummer100k_f32 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 600 python3 tools/shard_sim.py 2>&1 | grep -v amdgpu | tail -6
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
