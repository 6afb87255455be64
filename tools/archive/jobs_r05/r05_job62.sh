#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job62
mkdir -p $O
make -C examples > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for n in 100000 1000000 4000000; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -o p -- $ROOT/examples/leapfrog --nparts $n --steps 40 --warmup 5 > $O/prof_$n.log 2>&1
  f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n"; python3 - "$f" <<'PY' | tee $O/kernels_$n.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    name = r["Name"][:70]
    calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    if calls >= 40 and "k_list" not in name:
        print("%-72s calls/step %5.1f avg %8.1f us  per step %8.1f us" % (name, calls / 45.0, avg, calls / 45.0 * avg))
        tot += calls / 45.0 * avg
print("sum per step %.1f us" % tot)
PY
  rm -rf $O/prof_$n
done
