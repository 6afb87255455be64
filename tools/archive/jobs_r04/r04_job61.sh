#!/bin/bash
# Native leapfrog harness after the three-wave workgroups and the first-call launch order; then the whole suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
make -C examples > /dev/null 2>&1
for n in 100000 350000 1000000 4000000; do
  for v in 1 0; do echo "RK_FIRST_ORDER=$v $(RK_FIRST_ORDER=$v timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | cut -c100-330)"; done
done
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
