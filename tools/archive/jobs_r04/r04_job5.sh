#!/bin/bash
# k_common: chunk size (targets per wavefront) A/B at 4M; serial class kernels for the k_common duration, then the default run.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job5
mkdir -p $O
summ() { python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s value %.1f ms_per_step %.4f kernel_ms %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms")))
' $1 "$2" || tail -3 ${1%.json}.err; }
RK_COMMON=0 timeout 600 python3 bench.py --no-cpu-baseline > $O/b_members.json 2> $O/b_members.err; summ $O/b_members.json "members (RK_COMMON=0)"
for v in default c32 c64u2 c128 c256; do
  lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so; [ $v = default ] && lib=$ROOT/rakau_amd/lib/librakau_amd.so
  RAKAU_AMD_LIB=$lib RK_COMMON=1 timeout 600 python3 bench.py --no-cpu-baseline > $O/b_$v.json 2> $O/b_$v.err; summ $O/b_$v.json "k_common $v"
done
RK_COMMON=0 timeout 600 python3 bench.py --no-cpu-baseline > $O/b_members2.json 2> $O/b_members2.err; summ $O/b_members2.json "members (RK_COMMON=0)"
cd /tmp && export TMPDIR=/tmp
for v in default c128; do
  lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so; [ $v = default ] && lib=$ROOT/rakau_amd/lib/librakau_amd.so
  RAKAU_AMD_LIB=$lib RK_COMMON=1 RK_SERIAL_CLASSES=1 RK_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$v -- python3 $ROOT/bench.py --no-cpu-baseline > $O/serial_$v.log 2>&1
  f=$(find $O/serial_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v serial"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if "k_list" in n or "k_super" in n or "k_pc" in n or "k_common" in n:
        print("%-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  find $O/serial_$v -name "*kernel_trace.csv" -delete
done
