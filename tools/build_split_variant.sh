#!/bin/bash
# Library variant that differs from rakau_amd/lib only in rk_kernels_split.o (knobs of the split traversal):
# tools/build_split_variant.sh <name> [-DX=..]...  ->  rakau_amd/lib_<name>/librakau_amd.so  (select with RAKAU_AMD_LIB=)
name=$1; shift
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
d=../lib_$name; mkdir -p $d
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS "$@" -c rk_kernels_split.hip -o $d/rk_kernels_split.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd.so ../lib/rk_state.o ../lib/rk_launch.o ../lib/rk_host_out.o ../lib/rk_replica.o ../lib/rk_kernels.o ../lib/rk_kernels_list.o ../lib/rk_kernels_pc.o $d/rk_kernels_split.o ../lib/rk_build.o ../lib/rk_pool.o ../lib/rk_tree_capi.o -pthread -ldl
