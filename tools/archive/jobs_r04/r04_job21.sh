#!/bin/bash
# k_list_any (the one-launch kernel of calls over at most 60 000 critical nodes) compiled for 5 (default) / 6 / 7 waves per SIMD:
# the 8 equal-work shards of the 4M tree, 1M, 350k. Same box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job21
mkdir -p $O
for v in current any6 any7 current any6 any7; do
  lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
  echo "== $v shards: $(RAKAU_AMD_LIB=$lib timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -E 'N=8 work|N=4 work|N=2 work' | sed 's/variant 0 //' | tr '\n' ';')"
  for np in 350000 1000000; do
    RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --workload plummer100k_f32 --nparts $np --no-cpu-baseline > $O/b_${v}_$np.json 2> $O/b_${v}_$np.err
    python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   n=%s device-resident %.4f ms (kernel %.4f) seam %.4f" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"]))
' $O/b_${v}_$np.json $np
  done
done
