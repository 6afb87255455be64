#!/bin/bash
# Collects the round's judged evidence on the GPU box into gpurun_out/<tag>/ (copied into profiles/<round>/ afterwards):
#  * the whole -m gpu suite with durations;
#  * one bench line per BASELINE workload WITH the cpu_baseline / parity leg, the default line, the device-builder line,
#    the self-launched 2-rank rehearsal;
#  * rocprofv3 --kernel-trace --stats of the default bench command (class kernels overlapped, and back to back);
#  * PMC passes (separate --pmc runs, no tracing domains) and the HBM traffic of one step.
# usage (through gpurun): tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "${2:-}" != "prof-only" ]; then
( time timeout 1500 python3 -m pytest tests -m gpu -q --durations=12 ) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
for wl in plummer4m_f32_accpot plummer16m_f64 plummer64m_f32 plummer100k_f32; do
  timeout 900 python3 bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 600 python3 bench.py --builder device > $OUT/bench_device_builder.json 2> $OUT/bench_device_builder.err
RK_BENCH_SINGLE_DEVICE=1 RK_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 > $OUT/bench_selflaunch_2ranks_1gpu.json 2> $OUT/bench_selflaunch.err
timeout 900 python3 tools/shard_sim.py 4000000 > $OUT/shard_sim.txt 2>&1
fi
# Profiling runs: bench.py --no-pageable-leg, i.e. the timed steps only (the extra calls into pageable arrays run in two
# sub-range parts, whose launches would mix into the per-kernel averages).
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlapped -- python3 $ROOT/bench.py --no-cpu-baseline --no-pageable-leg > $OUT/bench_overlapped.log 2>&1
RK_SERIAL_CLASSES=1 RK_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $ROOT/bench.py --no-cpu-baseline --no-pageable-leg > $OUT/bench_serial.log 2>&1
BENCH_ARGS="" bash $ROOT/tools/prof_pmc.sh gpurun_out/$TAG/pmc > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
cd $ROOT
bash tools/measure_traffic.sh plummer4m_f32 > $OUT/traffic.log 2>&1
cp gpurun_out/traffic_plummer4m_f32.json $OUT/traffic.json 2>/dev/null
# One-launch kernels of the small workloads.
cd /tmp
for wl in 100k:plummer100k_f32:100000 1m:plummer100k_f32:1000000; do
  tag=${wl%%:*}; rest=${wl#*:}; key=${rest%%:*}; np=${rest#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small_$tag -- python3 $ROOT/bench.py --workload $key --nparts $np --no-cpu-baseline --no-pageable-leg > $OUT/bench_small_$tag.log 2>&1
done
cd $ROOT
# Kernel traces are large and not judged: keep the stats.
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_stats.csv" | head
