#!/bin/bash
# k_list class kernels: one / two / four list entries per wavefront (RK_NPW), same box, alternating. 4M.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job18
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s seam pinned %.4f ms (kernel %.4f) | device-resident %.4f (kernel %.4f) | pageable %.4f" % (sys.argv[2], d["ms_per_step"], d["kernel_ms"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_call_host_outputs_pageable"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2 3; do
  for v in current npw2 npw4; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "$v"
  done
done
