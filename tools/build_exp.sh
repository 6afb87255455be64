#!/bin/bash
# Round 5: experimental build of the traversal kernels with extra -D flags, everything else taken from the in-tree build:
#   tools/build_exp.sh <name> [-DX=..]...  ->  rakau_amd/lib_exp_<name>/librakau_amd.so (+ the cpu512 / xcheck libraries copied)
# Select with RAKAU_AMD_LIB=.../lib_exp_<name>/librakau_amd.so. Only rk_kernels_list.hip and rk_kernels_pc.hip are recompiled
# (knobs of rk_list_common.hpp / rk_device.hpp); knobs of rk_common.hpp need tools/build_variant_full.sh.
name=$1; shift
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
d=../lib_exp_$name; mkdir -p $d
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 --offload-compress -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS "$@" -c rk_kernels_list.hip -o $d/rk_kernels_list.o &
/opt/rocm/bin/hipcc $FLAGS "$@" -c rk_kernels_pc.hip -o $d/rk_kernels_pc.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd.so ../lib/rk_state.o ../lib/rk_launch.o ../lib/rk_host_out.o ../lib/rk_replica.o ../lib/rk_kernels.o $d/rk_kernels_list.o $d/rk_kernels_pc.o ../lib/rk_xcheck_loader.o ../lib/rk_build.o ../lib/rk_pool.o ../lib/rk_tree_capi.o -pthread -ldl || exit 1
cp ../lib/librakau_amd_cpu512.so ../lib/librakau_amd_xcheck.so $d/ 2>/dev/null
ls -la $d/librakau_amd.so
