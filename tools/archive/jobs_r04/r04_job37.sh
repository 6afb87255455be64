#!/bin/bash
# 4M fp32 accs_pots_u (four accumulators per target): the R = 4 (and R = 3) class kernels with one wave per SIMD fewer.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job37
mkdir -p $O
for rep in 1 2 3; do
  for v in current q2a q2b; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 900 python3 bench.py --workload plummer4m_f32_accpot --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
    python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s device-resident %.4f ms (kernel %.4f) seam %.4f (kernel %.4f) frac %.4f" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"], d["roofline"]["frac"]))
' $O/b_${v}_$rep.json $v || tail -3 $O/b_${v}_$rep.err
  done
done
