#!/bin/bash
# ONE launch over a merged light-tail list of all classes (k_list_any compiled for 6 waves per SIMD; round-3 experiment RK_ANY_TAIL=1 re-run on
# round 5's kernels) against the four class kernels: 4M / 2M device-resident kernel ms + hashes, three rounds; then the seam's call.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job23
mkdir -p $O
L=$ROOT/rakau_amd/lib_exp_anytail6/librakau_amd.so
for rep in 1 2 3; do
  RAKAU_AMD_LIB=$L RK_ANY_TAIL=0 timeout 600 python3 tools/pc_ring_probe.py 2000000,4000000 2>&1 | tail -1 | sed 's/^/class kernels  /' | tee -a $O/probe.txt
  RAKAU_AMD_LIB=$L RK_ANY_TAIL=1 timeout 600 python3 tools/pc_ring_probe.py 2000000,4000000 2>&1 | tail -1 | sed 's/^/one launch     /' | tee -a $O/probe.txt
done
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-14s value %.1f ms %.4f kernel_ms %s | device-resident %.1f ms %.4f kernel_ms %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["value_device_resident"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2; do
  for v in 0 1; do
    RAKAU_AMD_LIB=$L RK_ANY_TAIL=$v timeout 600 python3 bench.py --no-cpu-baseline --no-pageable-leg > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "RK_ANY_TAIL=$v" | tee -a $O/bench.txt
  done
done
