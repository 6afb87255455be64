#!/bin/bash
# Shader clock the chip holds while the 4M step runs: scalar against packed bodies (-DRK_TRACE builds, s_memtime / s_memrealtime
# per wave).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job3
mkdir -p $O
for rep in 1 2; do
for v in trace trace_pk; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so VARIANT=2 RK_GRAPH=0 timeout 300 python3 tools/trace_waves.py $O/tr_$v.npz 4000000 > $O/tr_$v.log 2>&1
  echo "== $v"; tail -1 $O/tr_$v.log; python3 tools/trace_digest.py $O/tr_$v.npz > $O/digest_${v}_$rep.txt 2>&1; head -4 $O/digest_${v}_$rep.txt; grep "^R " $O/digest_${v}_$rep.txt | head -4
done; done
rm -f $O/*.npz
