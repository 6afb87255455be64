#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for mx in 30000 60000 100000 1000000; do
  for n in 1000000 2000000 4000000; do
    echo "MAX=$mx leapfrog n=$n: $(RK_ANY_FIRST_MAX=$mx timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | grep -o '"ms_per_step.*ms_traversal": [0-9.]*')"
  done
done
