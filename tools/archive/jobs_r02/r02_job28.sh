#!/bin/bash
# Wave timelines of a 0.5M-particle shard of the 4M tree under the launch-plan / priority knobs.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job28
mkdir -p $OUT
cd $ROOT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  tag=k$1_p$2
  RK_PLAN_K=$1 RK_PRIO=$2 python3 tools/trace_waves.py $OUT/trace_$tag.npz 4000000 0.0 0.125 > $OUT/trace_$tag.log 2>&1
  echo "== RK_PLAN_K=$1 RK_PRIO=$2" >> $OUT/digest.txt
  python3 tools/trace_digest.py $OUT/trace_$tag.npz >> $OUT/digest.txt 2>&1
done
RK_PLAN_K=1 RK_PRIO=1 python3 tools/trace_waves.py $OUT/trace_100k_k1_p1.npz 100000 > $OUT/trace_100k.log 2>&1
echo "== 100k RK_PLAN_K=1 RK_PRIO=1" >> $OUT/digest.txt
python3 tools/trace_digest.py $OUT/trace_100k_k1_p1.npz >> $OUT/digest.txt 2>&1
cat $OUT/digest.txt
