#!/bin/bash
# Full GPU suite on the build with 19 knobs removed + the device-made light-tail arrangement; default bench line; smoke.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job9
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python3 bench.py 2>&1 | tail -1 | tee $O/bench_4m.json
