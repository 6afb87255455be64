#!/bin/bash
# Leaf gathering with lane = particle (RK_LEAF_LANE_PARTICLE=1) on top of switched-off idle lanes (RK_MASK_IDLE=1): kernel ms + result
# hashes at several sizes, then the default bench line, alternating on one box. base = round 4's kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job6
mkdir -p $O
for rep in 1 2; do
  for v in base mask_nolp mask_lp; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 100000,350000,500000,1000000,2000000,4000000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s value %.1f ms %.4f kernel_ms %s | device-resident %.1f ms %.4f kernel_ms %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms"], d["value_device_resident"], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for rep in 1 2; do
  for v in base mask_nolp mask_lp; do
    lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --no-cpu-baseline > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "$v" | tee -a $O/bench.txt
  done
done
