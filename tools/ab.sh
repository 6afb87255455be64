#!/bin/bash
# A/B of experimental library builds on the GPU box: tools/ab.sh <name>... (name "base" = the in-tree library).
# Prints Mparticles/s and kernel ms of the default bench for each, interleaved twice.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
  RAKAU_AMD_LIB=$lib python3 $ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 $BENCH_ARGS 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_ms'])"
done; done
