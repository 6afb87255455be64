#!/bin/bash
# k_super + k_common (pure dense): probe, bench A/B, serial per-kernel durations.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job3
mkdir -p $O
timeout 900 python3 tools/common_eval_probe.py > $O/probe.txt 2>&1; echo "probe rc=$?"; grep -c " ok" $O/probe.txt; grep "list != pc" $O/probe.txt | head -3; tail -1 $O/probe.txt
for mode in 0 1 0 1; do
  RK_COMMON=$mode timeout 600 python3 bench.py --no-cpu-baseline > $O/bench_common$mode.json 2> $O/bench_common$mode.err
  python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("RK_COMMON=%s value %.1f ms_per_step %.4f kernel_ms %s frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["roofline"]["frac"]))
' $O/bench_common$mode.json $mode || tail -5 $O/bench_common$mode.err
done
cd /tmp && export TMPDIR=/tmp
for mode in 1; do
  RK_COMMON=$mode RK_SERIAL_CLASSES=1 RK_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial$mode -- python3 $ROOT/bench.py --no-cpu-baseline > $O/serial$mode.log 2>&1
  f=$(find $O/serial$mode -name "*kernel_stats.csv" | head -1)
  echo "== RK_COMMON=$mode serial"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if "k_list" in n or "k_super" in n or "k_pc" in n or "k_common" in n:
        print("%-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
