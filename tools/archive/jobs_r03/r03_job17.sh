#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job17; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_multidevice.py tests/test_cpp_header.py tests/test_gpu_reference_tests.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python3 -c "
import json; d=json.load(open('$OUT/bench_default.json')); print({k: d['host'].get(k) for k in ('rk_init_s','state_create_cold_s','first_call_ms')}, d['value'], d['kernel_ms'], d['roofline']['frac'], d.get('value_host_outputs'), d.get('ms_per_call_host_outputs'), d.get('value_host_outputs_pinned'))"
