#!/bin/bash
# Epilogue values of the list kernels derived again behind list building instead of spilled (RK_EPILOGUE_REMAT, in-tree = 1) against
# the previous build (lib_exp_noremat): default bench A/B, small launches, shards, and the HBM traffic of a step.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job16
mkdir -p $O
export BENCH_ARGS="--no-pageable-leg"
timeout 1200 tools/ab.sh exp_noremat base exp_noremat base 2>&1 | grep -v amdgpu.ids | tee $O/ab_4m.txt
for rep in 1 2; do
  for v in exp_noremat base; do
    if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
    RAKAU_AMD_LIB=$lib timeout 600 python3 tools/pc_ring_probe.py 100000,350000,1000000,2000000,4000000 2>&1 | tail -1 | tee -a $O/probe.txt
  done
done
timeout 900 python3 -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_call_caches.py -x -q 2>&1 | tail -2 | tee $O/tests.txt
bash tools/measure_traffic.sh plummer4m_f32 > $O/traffic.log 2>&1
cp gpurun_out/traffic_plummer4m_f32.json $O/traffic.json
python3 -c "
import json;d=json.load(open('$O/traffic.json'));print('traffic GB', d['hbm_bytes_per_step']/1e9, 'fetch KiB', d['fetch_size_kib_per_step'], 'write KiB', d['write_size_kib_per_step'])" | tee $O/traffic.txt
