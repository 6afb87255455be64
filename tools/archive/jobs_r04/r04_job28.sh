#!/bin/bash
# RK_COMPACT_POP: list-building batches that fill their 64 candidate lanes. Parity first, then speed at every size.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job28
mkdir -p $O
X=$ROOT/rakau_amd/lib_exp_cpop/librakau_amd.so
RAKAU_AMD_LIB=$X timeout 1500 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_config1_100k.py tests/test_gpu_quadtree.py tests/test_gpu_call_caches.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -4
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s device-resident %.4f ms (kernel %.4f) | seam %.4f (kernel %.4f)" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for np in 100000 1000000 4000000; do
  for rep in 1 2; do
    for v in current cpop; do
      lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$X
      RAKAU_AMD_LIB=$lib timeout 600 python3 bench.py --workload plummer100k_f32 --nparts $np --no-cpu-baseline > $O/b_${np}_${v}_$rep.json 2> $O/b_${np}_${v}_$rep.err; summ $O/b_${np}_${v}_$rep.json "n=$np $v"
    done
  done
done
for v in current cpop; do
  lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != current ] && lib=$X
  echo "== $v shards: $(RAKAU_AMD_LIB=$lib timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -E 'N=8 work|N=4 work|N=2 work' | sed 's/variant 0 //' | tr '\n' ';')"
done
