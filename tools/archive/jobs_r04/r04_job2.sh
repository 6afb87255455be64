#!/bin/bash
# Kernel durations with the class kernels back to back (RK_SERIAL_CLASSES=1), common lists evaluated by the members / by the pre-pass.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  RK_COMMON=$mode RK_SERIAL_CLASSES=1 RK_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial$mode -- python3 $ROOT/bench.py --no-cpu-baseline > $O/serial$mode.log 2>&1
  f=$(find $O/serial$mode -name "*kernel_stats.csv" | head -1)
  echo "== RK_COMMON=$mode serial"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if "k_list" in n or "k_super" in n or "k_pc" in n:
        print("%-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
