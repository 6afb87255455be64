#!/bin/bash
# Diagnostic builds of librakau_amd.so: lib_ablate (dense phase removed: list building only) and
# lib_stamps (in-kernel section timers). Select with RAKAU_AMD_LIB=<path>.
cd "$(dirname "$0")/../rakau_amd/csrc" || exit 1
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden"
for v in ablate:RK_ABLATE_DENSE stamps:RK_STAMPS; do
  d=../lib_${v%%:*}; m=${v##*:}; mkdir -p $d
  for f in rk_state rk_launch rk_host_out rk_replica rk_kernels rk_kernels_list rk_kernels_pc rk_kernels_split rk_build rk_pool; do /opt/rocm/bin/hipcc $FLAGS -D$m -c $f.hip -o $d/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librakau_amd.so $d/rk_state.o $d/rk_launch.o $d/rk_host_out.o $d/rk_replica.o $d/rk_kernels.o $d/rk_kernels_list.o $d/rk_kernels_pc.o $d/rk_kernels_split.o $d/rk_build.o $d/rk_pool.o ../lib/rk_tree_capi.o -pthread
done
