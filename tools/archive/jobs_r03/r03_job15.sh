#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job15; mkdir -p $OUT
for r in 0 1 0 1; do RK_HOST_REGISTER=$r timeout 200 python3 tools/host_register_probe.py 2>&1 | tail -1 | tee -a $OUT/host_register.txt; done
timeout 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default.json; python3 -c "
import json; d=json.load(open('$OUT/bench_default.json')); print({k: d['host'].get(k) for k in ('rk_init_s','state_create_cold_s','first_call_ms','upload_s')}, d['value'], d['kernel_ms'], d['roofline']['frac'], d.get('value_host_outputs'))"
