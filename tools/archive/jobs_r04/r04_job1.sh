#!/bin/bash
# First contact of round 4: CUDA-seam driver, the common-eval probe, 4M bench with the common lists evaluated by the members
# (round 3's path) and by the pre-pass.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r04_job1
O=gpurun_out/r04_job1
RK_ALIAS_DEVICES=4 timeout 600 tests/build/cuda_bridge_driver > $O/cuda_bridge.txt 2>&1; echo "cuda bridge rc=$?"; tail -3 $O/cuda_bridge.txt
timeout 900 python3 tools/common_eval_probe.py > $O/probe.txt 2>&1; echo "probe rc=$?"; grep -c ok $O/probe.txt; grep FAIL $O/probe.txt | head; tail -2 $O/probe.txt
for mode in 0 1 0 1; do
  RK_COMMON=$mode timeout 600 python3 bench.py --no-cpu-baseline > $O/bench_common$mode.json 2> $O/bench_common$mode.err
  python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("RK_COMMON=%s value %.1f ms_per_step %.4f kernel_ms %s frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["roofline"]["frac"]))
' $O/bench_common$mode.json $mode || tail -5 $O/bench_common$mode.err
done
