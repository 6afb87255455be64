#!/bin/bash
# Experimental library with rk_build.hip taken from a git revision, everything else from the in-tree build (same-box A/B of the
# device tree build): tools/build_exp_rev.sh <name> <rev>  ->  rakau_amd/lib_exp_<name>/librakau_amd.so
name=$1; rev=$2
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
d=../lib_exp_$name; mkdir -p $d
git show $rev:rakau_amd/csrc/rk_build.hip > rk_build_rev_tmp.hip || exit 1
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 --offload-compress -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS -c rk_build_rev_tmp.hip -o $d/rk_build.o || exit 1
rm -f rk_build_rev_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd.so ../lib/rk_state.o ../lib/rk_launch.o ../lib/rk_host_out.o ../lib/rk_replica.o ../lib/rk_kernels.o ../lib/rk_kernels_list.o ../lib/rk_kernels_pc.o ../lib/rk_xcheck_loader.o $d/rk_build.o ../lib/rk_pool.o ../lib/rk_tree_capi.o -pthread -ldl || exit 1
cp ../lib/librakau_amd_cpu512.so ../lib/librakau_amd_xcheck.so $d/ 2>/dev/null
ls -la $d/librakau_amd.so
