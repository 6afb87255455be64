#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_leapfrog.py tests/test_gpu_quadtree.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for n in 100000 1000000 4000000; do
  echo "leapfrog n=$n: $(timeout 300 examples/leapfrog --nparts $n --steps 60 --warmup 5 2>&1 | tail -1 | grep -o '"ms_per_step.*ms_traversal": [0-9.]*')"
done
