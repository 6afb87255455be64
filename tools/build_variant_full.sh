#!/bin/bash
# Experimental build of the whole library with extra -D flags (for knobs in rk_common.hpp):
# tools/build_variant_full.sh <name> [-DX=..]...  ->  rakau_amd/lib_<name>/librakau_amd.so
name=$1; shift
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
d=../lib_$name; mkdir -p $d
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"
for f in rk_state rk_launch rk_host_out rk_replica rk_kernels rk_kernels_list rk_kernels_pc rk_xcheck_loader rk_kernels_xcheck rk_kernels_split rk_build rk_pool; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o $d/$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd.so $d/rk_state.o $d/rk_launch.o $d/rk_host_out.o $d/rk_replica.o $d/rk_kernels.o $d/rk_kernels_list.o $d/rk_kernels_pc.o $d/rk_xcheck_loader.o $d/rk_build.o $d/rk_pool.o ../lib/rk_tree_capi.o -pthread -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd_xcheck.so $d/rk_kernels_xcheck.o $d/rk_kernels_split.o -pthread
cp ../lib/librakau_amd_cpu512.so $d/ 2>/dev/null
