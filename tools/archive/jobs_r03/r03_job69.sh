#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for k in 16 8 32 64; do
  echo "RK_SUPER_K=$k first calls: $(RK_SUPER_K=$k timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1 | sed 's/rebuild+traversal [0-9.]* [0-9a-f]*//g')"
  echo "RK_SUPER_K=$k bench 100k: $(RK_SUPER_K=$k timeout 300 python3 bench.py --workload plummer100k_f32 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  RK_SUPER_K=$k RK_SUPER_CACHE=0 timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -E "N=8 work"
done
