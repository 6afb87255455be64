#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
for wl in plummer100k_f32 plummer4m_f32; do
timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-pageable-leg 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$wl', d['value'], d['ms_per_step'], d['kernel_ms'], 'device-resident', d['value_device_resident'], d['ms_per_step_device_resident'], d['kernel_ms_device_resident'])"
done
timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -v amdgpu | tail -2
