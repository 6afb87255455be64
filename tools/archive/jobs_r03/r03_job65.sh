#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for lib in lib_base lib; do
  export LD_LIBRARY_PATH=$ROOT/rakau_amd/$lib:${LD_LIBRARY_PATH:-}
  for n in 100000 1000000; do
    echo "$lib leapfrog n=$n: $(LD_PRELOAD=$ROOT/rakau_amd/$lib/librakau_amd.so timeout 300 examples/leapfrog --nparts $n --steps 60 --warmup 5 2>&1 | tail -1 | grep -o '"ms_per_step.*ms_traversal": [0-9.]*')"
  done
done
done
