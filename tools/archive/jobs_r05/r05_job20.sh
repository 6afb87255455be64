#!/bin/bash
# Whole GPU suite, the default bench line, the leapfrog harness: state of the tree after the kernel, build and stream changes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job20
mkdir -p $O
( time timeout 2300 python3 -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee $O/pytest.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
make -C examples > /dev/null 2>&1
for n in 100000 350000 1000000 2000000 4000000; do
  timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | cut -c100-330 | tee -a $O/leapfrog.txt
done
