#!/bin/bash
# After the library split (cross-check kernels in librakau_amd_xcheck.so, MAC read at run time): the whole GPU suite, the default
# bench line (value = the seam's call), 100k.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r04_job6
mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-700 $O/bench_default.json; tail -2 $O/bench_default.err
python3 bench.py --workload plummer100k_f32 --no-cpu-baseline > $O/bench_100k.json 2> $O/bench_100k.err; cut -c1-400 $O/bench_100k.json
