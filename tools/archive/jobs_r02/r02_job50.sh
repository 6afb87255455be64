#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job50
mkdir -p $OUT
cd $ROOT
( timeout 1500 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_reference_tests.py tests/test_gpu_full_size.py tests/test_gpu_quadtree.py tests/test_gpu_leapfrog.py tests/test_cpp_header.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log | cut -c1-300
for i in 1 2; do
timeout 900 python3 bench.py --workload plummer16m_f64 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('16M fp64:', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['cpu_baseline'].get('parity_max_rel_err'), d['cpu_baseline'].get('parity_median_rel_err'))" | tee -a $OUT/bench.txt
done
