#!/bin/bash
# A caller that re-creates its state every step (the reference's rocm_state after update_particles): this round against round 3.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "== round 4"; timeout 600 python3 tools/seam_step_probe.py 4000000 2>&1 | grep -v amdgpu
echo "== round 3 library"; RAKAU_AMD_LIB=$ROOT/rakau_amd/lib_exp_r03/librakau_amd.so timeout 600 python3 tools/seam_step_probe.py 4000000 2>&1 | grep -v amdgpu
tests/build/cuda_bridge_driver timing 4000000 2>&1 | grep -v amdgpu.ids
timeout 900 python3 -m pytest tests/test_gpu_host_outputs.py tests/test_gpu_multidevice.py tests/test_gpu_state_create.py -m gpu -x -q 2>&1 | tail -3
