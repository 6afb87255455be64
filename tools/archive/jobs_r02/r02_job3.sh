#!/bin/bash
# Round 2, GPU call 3: first run of the producer / consumer kernel: correctness vs the list kernel, times at several sizes
# and launch bounds, strong-scaling rehearsal per variant.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job3
mkdir -p $OUT
cd $ROOT
timeout 600 python3 tools/pc_check.py > $OUT/pc_check_w8.txt 2>&1; cat $OUT/pc_check_w8.txt | grep -v amdgpu.ids
for v in pc7 pc6; do
  RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_$v/librakau_amd.so timeout 600 python3 tools/pc_check.py 100000 4000000 > $OUT/pc_check_$v.txt 2>&1; echo "== $v"; grep -v amdgpu.ids $OUT/pc_check_$v.txt
done
timeout 900 python3 tools/shard_sim.py 4000000 2,3,4 > $OUT/shard_sim.txt 2>&1; grep -v amdgpu.ids $OUT/shard_sim.txt
( timeout 1200 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_reference_tests.py tests/test_gpu_quadtree.py tests/test_cpp_header.py tests/test_gpu_leapfrog.py -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
