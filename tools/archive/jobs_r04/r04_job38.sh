#!/bin/bash
# Sources in flight per lane in the dense loop (RK_UNR1..4), one knob at a time against the tree's build. 4M fp32, same box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job38
mkdir -p $O
for rep in 1 2; do
  for v in current u1_6 u1_2 u2_3 u3_2 u4_2; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 900 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
    python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s device-resident %.4f ms (kernel %.4f) seam %.4f (kernel %.4f)" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"]))
' $O/b_${v}_$rep.json $v || tail -3 $O/b_${v}_$rep.err
  done
done
