#!/bin/bash
# Small repeated calls: one k_list_any launch (default) against two launches by register budget (RK_ANY=5: R >= 3 at 5 waves per
# SIMD first, R <= 2 at 7 waves per SIMD). Shards of the 4M tree, 350k, 1M. Bits must agree.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job33
mkdir -p $O
for rep in 1 2; do
for any in default 5; do
  ex=""; [ $any != default ] && ex="RK_ANY=$any"
  echo "== RK_ANY=$any shards: $(env $ex timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -E 'N=8 work|N=4 work|N=2 work' | sed 's/variant 0 //' | tr '\n' ';')"
  for np in 350000 1000000; do
    env $ex timeout 600 python3 bench.py --workload plummer100k_f32 --nparts $np --no-cpu-baseline > $O/b_${any}_$np.json 2> $O/b_${any}_$np.err
    python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   n=%s device-resident %.4f ms (kernel %.4f) seam %.4f (kernel %.4f)" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"]))
' $O/b_${any}_$np.json $np
  done
done
done
RK_ANY=5 timeout 900 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
