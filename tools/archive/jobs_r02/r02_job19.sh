#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job19
mkdir -p $OUT
cd $ROOT
for lib in lib lib_sig1; do
  export RAKAU_AMD_LIB=$ROOT/rakau_amd/$lib/librakau_amd.so
  echo "== $lib"; timeout 300 python3 tools/host_timing.py 2>&1 | grep "^call" | tr '\n' ' '; echo
  timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench: value', d['value'], 'kernel_ms', d['kernel_ms'], 'host outputs', d.get('value_host_outputs'), d.get('ms_per_call_host_outputs'))"
  ( RK_HOST_POISON=1 timeout 1200 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_reference_tests.py tests/test_cpp_header.py -m gpu -x -q ) > $OUT/pytest_$lib.log 2>&1; tail -3 $OUT/pytest_$lib.log | cut -c1-200
done
