#!/bin/bash
# Experimental build of librakau_amd.so with extra -D flags: tools/build_variant.sh <name> [-DX=..]...
# -> rakau_amd/lib_<name>/librakau_amd.so (select with RAKAU_AMD_LIB). Only the traversal kernels (rk_kernels_list.hip,
# rk_kernels_pc.hip) are recompiled.
name=$1; shift
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
d=../lib_$name; mkdir -p $d
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS "$@" -c rk_kernels_list.hip -o $d/rk_kernels_list.o &
/opt/rocm/bin/hipcc $FLAGS "$@" -c rk_kernels_pc.hip -o $d/rk_kernels_pc.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd.so ../lib/rk_state.o ../lib/rk_launch.o ../lib/rk_host_out.o ../lib/rk_replica.o ../lib/rk_kernels.o $d/rk_kernels_list.o $d/rk_kernels_pc.o ../lib/rk_build.o ../lib/rk_pool.o ../lib/rk_tree_capi.o -pthread
