#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in lib lib_exp_nosteal; do
  echo "$v RK_ANY_PERSIST=1 $(RAKAU_AMD_LIB=$ROOT/rakau_amd/$v/librakau_amd.so RK_ANY_PERSIST=1 timeout 300 python3 tools/pc_ring_probe.py 250000,500000,1000000 2>&1 | grep -v amdgpu | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_nosteal/librakau_amd.so RK_ANY_PERSIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_job55 -- python3 $ROOT/tools/pc_ring_probe.py 500000 > $ROOT/gpurun_out/r04_job55.log 2>&1
head -4 $ROOT/gpurun_out/r04_job55/*/*kernel_stats.csv | cut -c1-200
