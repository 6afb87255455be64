#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_job6; mkdir -p $OUT
export RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_trace/librakau_amd.so
RK_SL_PARTS_BELOW=0 VARIANT=4 RK_GRAPH=0 timeout 300 python3 tools/trace_waves.py $OUT/tr_v4_4m.npz 4000000 > $OUT/tr_v4_4m.log 2>&1
python3 tools/trace_digest.py $OUT/tr_v4_4m.npz > $OUT/trace_v4_4m.txt 2>&1; head -30 $OUT/trace_v4_4m.txt
RK_SL_PARTS_BELOW=0 VARIANT=4 RK_GRAPH=0 RK_SERIAL_CLASSES=1 timeout 300 python3 tools/trace_waves.py $OUT/tr_v4_4m_serial.npz 4000000 > $OUT/tr_v4_4m_serial.log 2>&1
python3 tools/trace_digest.py $OUT/tr_v4_4m_serial.npz > $OUT/trace_v4_4m_serial.txt 2>&1; head -30 $OUT/trace_v4_4m_serial.txt
rm -f $OUT/*.npz
