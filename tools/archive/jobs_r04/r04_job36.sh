#!/bin/bash
# 16M fp64 theta 0.5: the R = 4 (and R = 3) class kernels compiled for 3 waves per SIMD (no spills) against 4 (80 / 20 B of scratch).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job36
mkdir -p $O
for rep in 1 2; do
  for v in current f64r4w3 f64r34w3; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 900 python3 bench.py --workload plummer16m_f64 --steps 8 --warmup 2 --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
    python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s device-resident %.3f ms (kernel %.3f) seam %.3f (kernel %.3f) frac %.4f" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"]))
' $O/b_${v}_$rep.json $v || tail -3 $O/b_${v}_$rep.err
  done
done
