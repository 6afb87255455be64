#!/bin/bash
# Eager delivery of staged results (marker words + polling host threads) for rk_acc_pot() into pageable arrays: tests, then the bench
# line (value = pinned arrays, value_host_outputs_pageable beside it), three rounds.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job16
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_cpp_header.py tests/test_gpu_reference_tests.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for rep in 1 2 3; do
  timeout 600 python3 bench.py --no-cpu-baseline > $O/b_$rep.json 2> $O/b_$rep.err
  python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h=d.get("host",{})
print("value %.1f ms %.4f | device-resident %.1f | pageable %s ms=%s" % (d["value"], d["ms_per_step"], d["value_device_resident"], d.get("value_host_outputs_pageable"), d.get("ms_per_call_host_outputs_pageable", h.get("acc_pot_host_outputs_ms"))))
' $O/b_$rep.json || tail -3 $O/b_$rep.err
done
python3 tools/host_split_probe.py 4000000 0 2>&1 | tail -3
python3 tools/host_split_probe.py 2000000 0 2>&1 | tail -3
python3 tools/host_split_probe.py 4000000 2 2>&1 | tail -3
