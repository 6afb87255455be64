#!/bin/bash
# Instruction budget of the 4M step: VALU wave-instructions (PMC) of the ablation builds, the dynamic event counts (-DRK_COUNTS),
# and the cost of recording an event per call (microbenchmark).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job5
mkdir -p $O
timeout 120 tools/ubench/build/event_cost 2>&1 | tee $O/ubench_event_cost.txt
RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_counts/librakau_amd.so timeout 300 python3 tools/counts_probe.py 4000000 2>&1 | tee $O/counts_4m.txt
cd /tmp && export TMPDIR=/tmp
for v in base mask ab_nodense ab_nodense_noexact ab_nodense_noleaves ab_nodense_nocommon ab_nodense_nolist ab_norem ab_nolist; do
  lib=$ROOT/rakau_amd/lib/librakau_amd.so; [ $v != base ] && lib=$ROOT/rakau_amd/lib_exp_$v/librakau_amd.so
  RAKAU_AMD_LIB=$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_$v -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pageable-leg > $O/pmc_$v.log 2>&1
  echo "== $v"; python3 $ROOT/tools/pmc_summary.py $O/pmc_$v > $O/pmc_$v.txt 2>&1; grep -A4 "k_list<float\|k_super" $O/pmc_$v.txt | grep "k_list\|k_super\|SQ_INSTS_VALU"
  rm -rf $O/pmc_$v
done
