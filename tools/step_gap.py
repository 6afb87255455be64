#!/usr/bin/env python3
"""Back-to-back accs_u() calls on the 4M tree: wall time per call against the HIP-event kernel time of a call, under the
launch-path knobs given in the environment (RK_SUPER_CACHE, RK_GRAPH, RK_EVENTS)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, rakau_amd
from bench import plummer_numpy
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4000000
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
stream = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    st.acc_pot_device(0, mv, ptrs, stream=stream)
torch.cuda.synchronize()
if os.environ.get("RK_TIMING") == "0":
    st.set_timing(False)
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    for _ in range(20):
        st.acc_pot_device(0, mv, ptrs, stream=stream)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
st.set_timing(True)
kms = []
for _ in range(8):
    st.acc_pot_device(0, mv, ptrs, stream=stream)
    kms.append(st.last_kernel_ms())
print("n=%d RK_SUPER_CACHE=%s RK_GRAPH=%s RK_TIMING=%s: ms per call back to back %.4f, kernel ms (events) %.4f" % (
    n, os.environ.get("RK_SUPER_CACHE", "1"), os.environ.get("RK_GRAPH", "1"), os.environ.get("RK_TIMING", "1"), best, float(np.median(kms))))
