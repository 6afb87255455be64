#!/bin/bash
# Round 3, GPU call 3: split traversal with parts; per-kernel times at 100k and 4M (rocprofv3 csv).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job3
mkdir -p $OUT
cd $ROOT
SIZES=1e5,1e6,4e6 timeout 600 python3 tools/split_check.py > $OUT/split_check.txt 2>&1
grep -E "FAIL|CHECK|kernel ms|Error|error" $OUT/split_check.txt | tail -30
cd /tmp && export TMPDIR=/tmp
for n in 100000 4000000; do
  RK_SERIAL_CLASSES=1 RK_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -o p -- python3 $ROOT/tools/run_variant.py $n 4 30 > $OUT/run_$n.txt 2>&1
  f=$(find $OUT/prof_$n -name "*kernel_stats.csv" | head -1)
  echo "== n=$n serial classes"; grep "kernel ms" $OUT/run_$n.txt
  if [ -n "$f" ]; then head -9 "$f" | cut -d, -f1-8 | cut -c1-180; fi
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ov -o p -- python3 $ROOT/tools/run_variant.py 100000 4 30 > $OUT/run_ov.txt 2>&1
f=$(find $OUT/prof_ov -name "*kernel_stats.csv" | head -1); echo "== 100k overlapped + graph"; grep "kernel ms" $OUT/run_ov.txt; if [ -n "$f" ]; then head -9 "$f" | cut -d, -f1-8 | cut -c1-180; fi
find $OUT -name "*.csv" ! -name "*kernel_stats*" -delete; find $OUT -name "*.db" -delete
