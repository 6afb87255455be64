#!/bin/bash
# Final check of the round: the driver's own sequence (GPU suite, smoke, default bench line).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r04_job34
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_job34/pytest.log 2>&1; tail -1 gpurun_out/r04_job34/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
python3 bench.py > gpurun_out/r04_job34/bench.json 2> gpurun_out/r04_job34/bench.err; python3 -c '
import json
d=json.loads(open("gpurun_out/r04_job34/bench.json").read().strip().splitlines()[-1])
print("value %.1f ms %.4f kernel %.4f frac %.4f | device %.1f | pageable %.1f | cpu %.1f parity %.2e" % (d["value"], d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"], d["value_device_resident"], d["value_host_outputs_pageable"], d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_max_rel_err"]))'
