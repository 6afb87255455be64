#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job57; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
timeout 600 python3 -m pytest tests/test_gpu_call_caches.py tests/test_gpu_parity_basic.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for f in 0 1; do
  echo "RK_FUSE_SUPER=$f first calls: $(RK_FUSE_SUPER=$f timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1)"
  echo "RK_FUSE_SUPER=$f bench 100k: $(RK_FUSE_SUPER=$f timeout 300 python3 bench.py --workload plummer100k_f32 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  RK_FUSE_SUPER=$f timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -E "full|N=8 work|N=4 work"
  echo "RK_FUSE_SUPER=$f RK_SUPER_CACHE=0 repeated: $(RK_FUSE_SUPER=$f RK_SUPER_CACHE=0 timeout 300 python3 tools/any_probe3.py 2>&1 | tail -1)"
done
