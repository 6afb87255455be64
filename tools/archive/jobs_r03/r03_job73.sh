#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
BENCH_ARGS="--workload plummer100k_f32" bash tools/prof_pmc.sh gpurun_out/r03_job73/pmc > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r03_job73/pmc > gpurun_out/r03_job73/pmc_summary.txt 2>&1
python3 tools/pmc_digest.py gpurun_out/r03_job73/pmc_summary.txt 2>&1 | tail -8 | cut -c1-230
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $ROOT/gpurun_out/r03_job73/ifetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload plummer100k_f32 > $ROOT/gpurun_out/r03_job73/ifetch.log 2>&1
f=$(find $ROOT/gpurun_out/r03_job73/ifetch -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
if len(sys.argv) < 2 or not sys.argv[1]:
    print("no counter file"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in acc.items():
    if "k_pc_any" in k or "k_list" in k:
        print(k, {a: "%.3g" % (b / max(cnt[k], 1)) for a, b in v.items()})
PY
