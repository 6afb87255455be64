#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r05_job22
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $ROOT/bench.py --no-cpu-baseline --no-pageable-leg --steps 6 --warmup 2 > $O/bench.log 2>&1
python3 $ROOT/tools/seam_timeline.py $O/tr | tee $O/timeline.txt
tail -1 $O/bench.log | cut -c1-300
rm -rf $O/tr
