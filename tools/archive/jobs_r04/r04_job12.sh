#!/bin/bash
# After taking k_common out and the MAC back to a template parameter: same-box A/B against round 3's library, then the GPU suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job12
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s value %.1f ms %.4f kernel_ms %s | device-resident %.1f ms %.4f kernel_ms %s | pageable %.1f ms %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["value_device_resident"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["value_host_outputs_pageable"], d["ms_per_call_host_outputs_pageable"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2; do
  for v in current r03; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "$v"
  done
done
( time timeout 1800 python3 -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
