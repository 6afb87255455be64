#!/usr/bin/env python3
"""First call on a tree (no launch plan, no graph: what every step of a time-stepping loop is): ms per call of
rebuild_device + acc_pot_device pairs and of the traversal alone, RK_ANY_FIRST=0 (class kernels forked onto side streams)
against the default (one launch over the class lists read backwards). Result bits hashed (must not differ)."""
import os, sys, hashlib, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
res = []
for n in (100_000, 350_000, 1_000_000, 1_800_000):
    m, x, y, z = plummer_numpy(n, "float32")
    ts = [torch.as_tensor(v).cuda() for v in (x, y, z, m)]
    ptrs_in = [t.data_ptr() for t in ts]
    st = rakau_amd.State.build_device(ptrs_in, n, np.float32)
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    st.set_timing(True)
    ms = []
    for _ in range(12):
        st.rebuild_device(ptrs_in)
        st.acc_pot_device(0, mv, ptrs)
        ms.append(st.last_kernel_ms())
    st.set_timing(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st.rebuild_device(ptrs_in)
        st.acc_pot_device(0, mv, ptrs)
    torch.cuda.synchronize()
    pair = (time.perf_counter() - t0) / 20 * 1e3
    h = hashlib.sha1()
    for o in outs:
        h.update(o.cpu().numpy().tobytes())
    res.append("%dk(%d nodes) traversal %.4f rebuild+traversal %.3f %s" % (n // 1000, st.n_crit, float(np.median(ms[3:])), pair, h.hexdigest()[:8]))
print("RK_ANY_FIRST=%s: %s" % (os.environ.get("RK_ANY_FIRST", "1"), "; ".join(res)))
