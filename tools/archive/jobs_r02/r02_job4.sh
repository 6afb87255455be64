#!/bin/bash
# Round 2, GPU call 4: producer / consumer kernel after the leaf-gather fix (correctness + times), the chunked
# kernel / device-to-host pipeline of rk_acc_pot (host-output rate), and the tests that cover both.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job4
mkdir -p $OUT
cd $ROOT
timeout 600 python3 tools/pc_check.py > $OUT/pc_check.txt 2>&1; grep -v amdgpu.ids $OUT/pc_check.txt
for ch in 1 2 4 8; do
  RK_HOST_CHUNKS=$ch timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks $ch: value', d['value'], 'kernel_ms', d['kernel_ms'], 'host outputs', d.get('value_host_outputs'), d.get('ms_per_call_host_outputs'))"
done
timeout 900 python3 tools/shard_sim.py 4000000 2,3,4 > $OUT/shard_sim.txt 2>&1; grep -v amdgpu.ids $OUT/shard_sim.txt
( timeout 1200 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_full_size.py tests/test_integration_bridge.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
