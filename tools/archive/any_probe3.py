#!/usr/bin/env python3
"""Kernel ms of repeated calls between 3.2k and 5k critical nodes (where the automatic variant used forked class launches on
the producer / consumer kernel, which are no longer replayed from a graph): RK_ANY=0 / 1 / 3 / unset."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
res = []
for n in (120_000, 140_000, 160_000, 180_000, 220_000):
    m, x, y, z = plummer_numpy(n, "float32")
    st = rakau_amd.Octree(x, y, z, m).state()
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(40):
        st.acc_pot_device(0, mv, ptrs)
    ms = []
    for _ in range(40):
        st.acc_pot_device(0, mv, ptrs)
        ms.append(st.last_kernel_ms())
    # wall clock of a queue of calls (launch overhead included)
    st.set_timing(False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        st.acc_pot_device(0, mv, ptrs)
    e1.record(); torch.cuda.synchronize()
    res.append("%dk(%d nodes) event %.4f queued %.4f" % (n // 1000, st.n_crit, float(np.median(ms[5:])), e0.elapsed_time(e1) / 200))
print("RK_ANY=%s: %s" % (os.environ.get("RK_ANY", "auto"), "; ".join(res)))
