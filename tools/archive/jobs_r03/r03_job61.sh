#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for lib in lib lib_pcw6; do
  for rep in 1 2; do
  echo "$lib: $(RAKAU_AMD_LIB=$ROOT/rakau_amd/$lib/librakau_amd.so timeout 300 python3 bench.py --workload plummer100k_f32 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["kernel_ms"])')"
  done
  echo "$lib: $(RAKAU_AMD_LIB=$ROOT/rakau_amd/$lib/librakau_amd.so timeout 300 python3 tools/any_probe3.py 2>&1 | tail -1 | cut -c1-200)"
done
