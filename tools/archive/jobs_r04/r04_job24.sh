#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_state_create.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -8
RK_BUILD_TIMING=1 RK_CREATE_ON_HOST=1 python3 - <<'PY' 2>&1 | grep -a "RK_BUILD_TIMING\|total" | tail -14
import time, numpy as np, rakau_amd
from bench import plummer_numpy
m, x, y, z = plummer_numpy(4_000_000, "float32")
t = rakau_amd.Octree(x, y, z, m)
p = t.p_its_u(); nodes = t.nodes()
for i in range(2):
    t0 = time.perf_counter(); s = rakau_amd.State(p[0], p[1], p[2], p[3], nodes, ncrit=128); print("total %.1f ms" % ((time.perf_counter()-t0)*1e3)); s.close()
PY
