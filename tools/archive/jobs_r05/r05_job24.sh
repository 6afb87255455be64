#!/bin/bash
# The seam's call (rk_acc_pot into pinned arrays) replayed from a hipGraph (RK_HOST_GRAPH=1, experiment) against direct launches: 4M and 100k.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job24
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s value %.1f ms %.4f kernel_ms %s | device-resident %.1f ms %.4f kernel_ms %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["value_device_resident"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2 3; do
  for v in 0 1; do
    RK_HOST_GRAPH=$v timeout 600 python3 bench.py --no-cpu-baseline --no-pageable-leg > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "4M RK_HOST_GRAPH=$v" | tee -a $O/bench.txt
  done
done
for rep in 1 2; do
  for v in 0 1; do
    RK_HOST_GRAPH=$v timeout 600 python3 bench.py --workload plummer100k_f32 --no-cpu-baseline --no-pageable-leg > $O/s_${v}_$rep.json 2> $O/s_${v}_$rep.err; summ $O/s_${v}_$rep.json "100k RK_HOST_GRAPH=$v" | tee -a $O/bench.txt
  done
done
