#!/bin/bash
# Pipelined upload of pageable arrays: native program (system HIP runtime) and Python (PyTorch's runtime), against RK_UPLOAD_PIPE=0.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for pipe in 1 0; do
  echo "== RK_UPLOAD_PIPE=$pipe native"; RK_UPLOAD_PIPE=$pipe tests/build/cuda_bridge_driver timing 4000000 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== RK_UPLOAD_PIPE=$pipe python"; RK_UPLOAD_PIPE=$pipe timeout 600 python3 tools/seam_step_probe.py 4000000 2>&1 | grep -v amdgpu | tail -4
done
timeout 900 python3 -m pytest tests/test_gpu_state_create.py tests/test_gpu_device_build.py tests/test_gpu_parity_basic.py -m gpu -x -q 2>&1 | tail -3
