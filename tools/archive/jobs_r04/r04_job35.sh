#!/bin/bash
# k_pc_any with the R = 1 nodes on one wavefront (list_node) instead of producer + consumer: 30k, 100k, 130k particles. Same bits.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job35
mkdir -p $O
X=$ROOT/rakau_amd/lib_exp_r1list/librakau_amd.so
for np in 30000 100000 130000; do
  for rep in 1 2; do
    for v in current r1list; do
      lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$X
      RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --workload plummer100k_f32 --nparts $np --no-cpu-baseline > $O/b_${np}_${v}_$rep.json 2> $O/b_${np}_${v}_$rep.err
      python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("n=%s %-8s device-resident %.4f ms (kernel %.4f) seam %.4f (kernel %.4f)" % (sys.argv[2], sys.argv[3], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"]))
' $O/b_${np}_${v}_$rep.json $np $v
    done
  done
done
RAKAU_AMD_LIB=$X timeout 900 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_config1_100k.py tests/test_gpu_parity_basic.py -m gpu -x -q 2>&1 | tail -3
