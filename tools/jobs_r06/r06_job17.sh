#!/bin/bash
# Final build of the round: the whole GPU suite, smoke, the default bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job17
mkdir -p $O
( time timeout 2400 python3 -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 600 python3 bench.py 2>&1 | tail -1 > $O/bench_4m.json
python3 -c "
import json;d=json.loads(open('$O/bench_4m.json').read());print(d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('value_device_resident'),d.get('value_host_outputs_pageable'))" | tee $O/bench_4m.txt
