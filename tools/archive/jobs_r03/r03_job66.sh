#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lf -- $ROOT/examples/leapfrog --nparts 100000 --steps 20 --warmup 5 > /tmp/lf.log 2>&1
tail -1 /tmp/lf.log | grep -o '"ms_per_step.*ms_traversal": [0-9.]*'
g=$(find /tmp/prof_lf -name "*kernel_trace.csv" | head -1)
python3 - "$g" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find one full step in the steady state: locate the k_maxabs / first build kernel occurrences
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_encode" in n or "k_codes" in n or "k_maxabs" in n]
starts = [i for i in idx if "k_maxabs" in names[i]] or idx
if len(starts) > 12:
    a, b = starts[-3], starts[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    busy = 0
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        print("%8.1f us  +%6.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
        prev_end = e
    print("step span %.1f us, kernels busy %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
PY
