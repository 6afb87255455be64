#!/bin/bash
# Small launches: the supergroup pre-pass on / off (RK_SUPER_K=0) at 100k, 350k, 1M; shard rehearsal of the 4M tree.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job17
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-34s device-resident %.4f ms (kernel %.4f) | seam pinned %.4f (kernel %.4f)" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"]))
' $1 "$2" || tail -3 ${1%.json}.err; }
for np in 100000 350000 1000000; do
  for k in 16 0 16 0; do
    RK_SUPER_K=$k timeout 600 python3 bench.py --workload plummer100k_f32 --nparts $np --no-cpu-baseline > $O/b_${np}_k$k.json 2> $O/b_${np}_k$k.err; summ $O/b_${np}_k$k.json "n=$np RK_SUPER_K=$k"
  done
done
timeout 900 python3 tools/shard_sim.py 4000000 2>&1 | tail -12
