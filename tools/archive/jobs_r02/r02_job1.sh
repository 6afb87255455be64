#!/bin/bash
# Round 2, GPU call 1: the whole -m gpu suite (with the new full-size tests for BASELINE configs 3/4/5), one bench
# line per BASELINE workload WITH the cpu_baseline / parity leg, the self-launched 2-rank rehearsal at 4M, and the
# kernel-trace + PMC passes of the default bench command.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job1
mkdir -p $OUT
cd $ROOT
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -E "avx|fma" | tr '\n' ' ' >> $OUT/host.txt; free -g >> $OUT/host.txt
( time timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for wl in plummer4m_f32 plummer4m_f32_accpot plummer16m_f64 plummer64m_f32 plummer100k_f32; do
  timeout 900 python3 bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -c 600 $OUT/bench_$wl.json; echo
done
RK_BENCH_SINGLE_DEVICE=1 RK_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 > $OUT/bench_selflaunch_2ranks_1gpu.json 2> $OUT/bench_selflaunch.err
tail -c 300 $OUT/bench_selflaunch_2ranks_1gpu.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlapped -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_overlapped.log 2>&1
BENCH_ARGS="" bash $ROOT/tools/prof_pmc.sh gpurun_out/r02_job1/pmc > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
head -50 $OUT/pmc_summary.txt
