#!/bin/bash
# Host-output calls on the state's own non-blocking stream with graph replay: tests that drive it (host outputs, several host
# threads / logical devices, bridges), then bench A/B against RK_HOST_GRAPH=0 at 4M and 100k.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job15
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_multidevice.py tests/test_integration_bridge.py tests/test_cpp_header.py tests/test_gpu_reference_tests.py tests/test_gpu_bench_multirank.py tests/test_gpu_parity_basic.py -m gpu -x -q 2>&1 | tail -4
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-26s seam pinned %.4f ms (kernel %.4f) | device-resident %.4f | pageable %.4f ms" % (sys.argv[2], d["ms_per_step"], d["kernel_ms"], d["ms_per_step_device_resident"], d["ms_per_call_host_outputs_pageable"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for wl in plummer4m_f32 plummer100k_f32; do
for rep in 1 2; do
  for g in 1 0; do
    RK_HOST_GRAPH=$g timeout 600 python3 bench.py --workload $wl --no-cpu-baseline > $O/b_${wl}_g${g}_$rep.json 2> $O/b_${wl}_g${g}_$rep.err; summ $O/b_${wl}_g${g}_$rep.json "$wl host graph $g"
  done
done
done
