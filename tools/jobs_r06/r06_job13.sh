#!/bin/bash
# Kernel timeline of one leapfrog step on the round 6 final build (rebuild part + traversal), 100k / 1M / 4M particles.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job13
mkdir -p $O
make -C examples > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for n in 100000 1000000 4000000; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$n -o p -- $ROOT/examples/leapfrog --nparts $n --steps 20 --warmup 5 > $O/prof_$n.log 2>&1
  echo "== $n"; python3 $ROOT/tools/rebuild_timeline.py $O/prof_$n 3 | tee $O/timeline_$n.txt
  rm -rf $O/prof_$n
done
