#!/bin/bash
# Completion of a blocking rk_acc_pot(): polling (RK_HOST_SPIN_US=3000, default) against the runtime's wait (0). 100k, 1M, 4M.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job20
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-26s seam pinned %.4f ms (kernel %.4f) | device-resident %.4f | pageable %.4f" % (sys.argv[2], d["ms_per_step"], d["kernel_ms"], d["ms_per_step_device_resident"], d["ms_per_call_host_outputs_pageable"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for np in 100000 1000000 4000000; do
  for rep in 1 2; do
    for spin in 3000 0; do
      RK_HOST_SPIN_US=$spin timeout 600 python3 bench.py --workload plummer100k_f32 --nparts $np --no-cpu-baseline > $O/b_${np}_s${spin}_$rep.json 2> $O/b_${np}_s${spin}_$rep.err; summ $O/b_${np}_s${spin}_$rep.json "n=$np spin=$spin"
    done
  done
done
