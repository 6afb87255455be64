#!/bin/bash
# A/B of environment knobs on the in-tree library: tools/ab_env.sh "RK_X=1" "RK_Y=2 RK_Z=3" ... ("-" = no knob)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e python3 $ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 $BENCH_ARGS 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['kernel_ms'])"
done; done
