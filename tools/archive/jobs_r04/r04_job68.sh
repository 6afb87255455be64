#!/bin/bash
# Why is the leapfrog harness's 4M traversal (1.95 ms) faster than the bench's repeated call (2.26)? Plan on/off in the bench,
# device builder, and the harness's own numbers.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for e in "" "RK_PLAN=0" "RK_GRAPH=0"; do
  echo "[$e] $(env $e timeout 300 python3 bench.py --no-cpu-baseline --no-pageable-leg 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'], 'dev', d['ms_per_step_device_resident'], d['kernel_ms_device_resident'], d['interactions_per_particle'])")"
done
echo "[builder device] $(timeout 300 python3 bench.py --no-cpu-baseline --no-pageable-leg --builder device 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'], 'dev', d['ms_per_step_device_resident'], d['kernel_ms_device_resident'], d['interactions_per_particle'])")"
make -C examples > /dev/null 2>&1
timeout 300 examples/leapfrog --nparts 4000000 --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-600
