#!/bin/bash
# On top of R <= 2 at 8 waves per SIMD: R = 3 at 7 (72 VGPRs), R = 4 at 6 (80), k_list_any at 6.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job12
mkdir -p $O
for rep in 1 2; do
  for v in w8s w8s_r3w7 w8s_r4w6 w8s_any6; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != none ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 100000,350000,500000,1000000,2000000,4000000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s value %.1f ms %.4f kernel_ms %s | device-resident %.1f ms %.4f kernel_ms %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["value_device_resident"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2; do
  for v in w8s w8s_r3w7 w8s_r4w6 w8s_any6; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != none ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "$v" | tee -a $O/bench.txt
  done
done
