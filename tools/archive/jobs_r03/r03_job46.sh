#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job46; mkdir -p $OUT
for rep in 1 2; do
for g in 1 0; do
  RK_GRAPH=$g timeout 600 python3 bench.py --no-cpu-baseline --steps 200 --warmup 30 > $OUT/b_$g.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$OUT/b_$g.json').read().strip().splitlines()[-1]); print('RK_GRAPH=$g', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
done
python3 bench.py --help | head -30
