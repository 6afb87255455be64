#!/bin/bash
# fp64: the R = 1 and R = 2 class kernels at 5 waves per SIMD (96 VGPRs, 24 / 36 bytes of scratch) against 4 (no scratch): 16M theta 0.5.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job34
mkdir -p $O
for rep in 1 2; do
  for v in base f64w5; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 900 python3 bench.py --workload plummer16m_f64 --steps 8 --warmup 2 --no-cpu-baseline --no-pageable-leg > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
    python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s device-resident %.3f ms (kernel %.3f) seam %.3f (kernel %.3f) frac %.4f" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"]))
' $O/b_${v}_$rep.json $v || tail -3 $O/b_${v}_$rep.err
  done
done
