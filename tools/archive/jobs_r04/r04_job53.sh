#!/bin/bash
# Stability of the suite with the three-wave producer / consumer workgroups: three runs in a row, then smoke and the leapfrog harness.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for i in 1 2 3; do timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2; done
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
make -C examples > /dev/null 2>&1; for n in 100000 1000000 4000000; do timeout 300 examples/leapfrog $n 2>&1 | tail -1; done
