#!/bin/bash
# Pageable host outputs: current library (stripped) / the same objects linked without -s / round 3's library, same box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job13
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s seam pinned %.4f ms | device-resident %.4f | pageable %.4f ms (kernel %.4f)" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_device_resident"], d["ms_per_call_host_outputs_pageable"], d["host"]["kernel_ms_host_outputs_pageable"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2; do
  for v in current nostrip r03; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "$v"
  done
done
