#!/bin/bash
# Experimental build with other knobs for k_common only (rk_kernels_common.hip: RK_SE_CHUNK, RK_SE_WPS, RK_SE_TILE, RK_SE_W32, RK_UNR4):
# tools/build_common_variant.sh <name> [-DX=..]...  ->  rakau_amd/lib_exp_<name>/librakau_amd.so (select with RAKAU_AMD_LIB=<path>)
name=$1; shift
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
d=../lib_exp_$name; mkdir -p $d
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS "$@" -c rk_kernels_common.hip -o $d/rk_kernels_common.o || exit 1
objs=$(ls ../lib/*.o | grep -v rk_kernels_common.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd.so $d/rk_kernels_common.o $objs -pthread -ldl
cp ../lib/librakau_amd_cpu512.so $d/ 2>/dev/null
ls -la $d/librakau_amd.so
