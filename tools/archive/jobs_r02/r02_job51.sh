#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job51
mkdir -p $OUT
cd $ROOT
for wl in plummer4m_f32_accpot plummer16m_f64 plummer64m_f32 plummer100k_f32; do
  timeout 900 python3 bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  cut -c1-200 $OUT/bench_$wl.json
done
timeout 600 python3 bench.py --builder device > $OUT/bench_device_builder.json 2> $OUT/bench_device_builder.err
RK_BENCH_SINGLE_DEVICE=1 RK_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 > $OUT/bench_selflaunch_2ranks_1gpu.json 2> $OUT/bench_selflaunch.err
