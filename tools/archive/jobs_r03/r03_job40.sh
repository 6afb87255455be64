#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job40; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 RK_BACKTRACE=1
MALLOC_MMAP_THRESHOLD_=33554432 timeout 200 python3 tools/stress_host_register.py 40 11 > $OUT/stress_heap.log 2>&1; echo "stress heap rc=$? $(tail -1 $OUT/stress_heap.log)"
timeout 200 python3 tools/stress_host_register.py 40 12 > $OUT/stress_mmap.log 2>&1; echo "stress mmap rc=$? $(tail -1 $OUT/stress_mmap.log)"
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d.get('host'), indent=None))"
for t in 1 8 16; do RK_HOST_THREADS=$t timeout 300 python3 tools/host_register_probe.py 2>&1 | tail -4; done
for i in 1 2 3; do
  timeout 900 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $OUT/run_$i.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then grep -n "native stack\|Error\|error" -B6 -A30 $OUT/run_$i.log | grep -v "^[0-9]*-  File" | head -80; fi
done
