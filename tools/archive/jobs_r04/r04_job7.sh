#!/bin/bash
# MAC read at run time (one code object for bh and bh_geom) against the MAC as a template parameter: same box, alternating.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job8
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s value %.1f ms %.4f kernel_ms %s | device-resident %.1f ms %.4f kernel_ms %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["value_device_resident"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2 3; do
  for v in runtime r03; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v = r03 ] && lib=$ROOT/rakau_amd/lib_exp_r03/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "MAC $v"
  done
done

