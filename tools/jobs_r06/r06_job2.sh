#!/bin/bash
# (a) do fp32 MFMA and fp32 VALU instructions of different waves on one SIMD run side by side? (tools/ubench/mfma_coexec.hip)
# (b) lane-derived addresses recomputed per batch instead of spilled (RK_RELAUNDER, in-tree = 1) against the previous build (= 0)
# (c) two-occupancy launch for repeated calls with a heavy-first plan (RK_ANY=4) against one k_list_any launch (auto)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job2
mkdir -p $O
timeout 300 tools/ubench/build/mfma_coexec 2>&1 | tee $O/ubench_mfma_coexec.txt
export BENCH_ARGS="--no-pageable-leg"
timeout 900 tools/ab.sh exp_nolaunder base 2>&1 | grep -v amdgpu.ids | tee $O/ab_4m.txt
for rep in 1 2; do
  for v in exp_nolaunder base; do
    if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 100000,350000,1000000,2000000 2>&1 | tail -1 | tee -a $O/probe_launder.txt
  done
done
for rep in 1 2; do
  for a in auto 4; do
    if [ $a = auto ]; then unset RK_ANY; else export RK_ANY=$a; fi
    timeout 600 python3 tools/pc_ring_probe.py 250000,350000,500000,750000,1000000,1500000 2>&1 | tail -1 | tee -a $O/probe_any.txt
  done
done
unset RK_ANY
for a in auto 4; do
  if [ $a = auto ]; then unset RK_ANY; else export RK_ANY=$a; fi
  echo "== RK_ANY=$a" | tee -a $O/shard_any.txt
  timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work\|full" | tee -a $O/shard_any.txt
done
unset RK_ANY
RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_nolaunder/librakau_amd.so timeout 600 python3 tools/shard_sim.py 2>&1 | grep "work\|full" | tee $O/shard_nolaunder.txt
