#!/bin/bash
# Round 6, first call: (a) the MFMA-accumulation microbenchmark, (b) the new tests (headline config on seven
# windows, sort paths), (c) this box's baseline of the default bench line and of the small launches.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job1
mkdir -p $O
timeout 300 tools/ubench/build/mfma_acc 2>&1 | tee $O/ubench_mfma_acc.txt
timeout 1500 python3 -m pytest tests/test_gpu_full_size.py::test_4m_accs_u_headline tests/test_gpu_leapfrog.py "tests/test_gpu_parity_basic.py::test_accs_pots_equal_accs_and_pots" -x -q 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python3 bench.py --no-pageable-leg 2>&1 | tail -1 | tee $O/bench_4m.json
timeout 600 python3 tools/shard_sim.py 2>&1 | tee $O/shard_sim.txt | tail -30
timeout 600 python3 tools/pc_ring_probe.py 100000,350000,1000000 2>&1 | tail -3 | tee $O/probe.txt
