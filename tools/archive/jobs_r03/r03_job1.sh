#!/bin/bash
# Round 3, GPU call 1: VALU issue rates with the sustained clock; first run of the split traversal (variant 4).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job1
mkdir -p $OUT
cd $ROOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > $OUT/ubench_valu_rates.txt 2>&1
tail -8 $OUT/ubench_valu_rates.txt
timeout 900 python3 tools/split_check.py > $OUT/split_check.txt 2>&1
grep -E "FAIL|CHECK|kernel ms|Error|error" $OUT/split_check.txt | tail -40
