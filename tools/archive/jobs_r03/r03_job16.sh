#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job16; mkdir -p $OUT
for r in 0 1 0 1; do RK_HOST_REGISTER=$r timeout 200 python3 tools/host_register_probe.py 2>&1 | tail -1 | tee -a $OUT/host_register.txt; done
for r in 0 1; do N=1e5 RK_HOST_REGISTER=$r timeout 200 python3 tools/host_register_probe.py 2>&1 | tail -1 | tee -a $OUT/host_register.txt; done
for r in 0 1; do N=64e6 RK_HOST_REGISTER=$r timeout 300 python3 tools/host_register_probe.py 2>&1 | tail -1 | tee -a $OUT/host_register.txt; done
