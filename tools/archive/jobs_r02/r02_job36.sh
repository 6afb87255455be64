#!/bin/bash
# What the per-call event records cost between back-to-back calls (experiment).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_job36
mkdir -p $OUT
cd $ROOT
cat > /tmp/evt.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, rakau_amd
from bench import plummer_numpy
n = 4000000
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
stream = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    st.acc_pot_device(0, mv, ptrs, stream=stream)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        st.acc_pot_device(0, mv, ptrs, stream=stream)
    torch.cuda.synchronize()
    print("RK_EVENTS=%s RK_GRAPH=%s ms/step %.4f" % (os.environ.get("RK_EVENTS", "2"), os.environ.get("RK_GRAPH", "1"), (time.perf_counter() - t0) / 20 * 1e3))
PY
for ev in 2 1 0; do for g in 1 0; do RK_EVENTS=$ev RK_GRAPH=$g python3 /tmp/evt.py 2>&1 | grep "ms/step" | tee -a $OUT/events.txt; done; done
