#!/usr/bin/env python3
"""ms per repeated full-range call over a range of tree sizes: queued back to back (as bench.py times them) and with a
synchronisation after every call. Looks for sizes where the launch sequence misbehaves."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, rakau_amd
from bench import plummer_numpy
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
sizes = [int(float(v)) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1.5e6,2e6,2.5e6,3e6,4e6,6e6".split(","))]
out = []
for n in sizes:
    m, x, y, z = plummer_numpy(n, "float32")
    st = rakau_amd.Octree(x, y, z, m).state()
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(40):
        st.acc_pot_device(0, mv, ptrs)
    st.set_timing(False); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        st.acc_pot_device(0, mv, ptrs)
    e1.record(); torch.cuda.synchronize()
    queued = e0.elapsed_time(e1) / 60
    st.set_timing(True)
    ms = []
    for _ in range(30):
        st.acc_pot_device(0, mv, ptrs); ms.append(st.last_kernel_ms())
    out.append("%.1fM(%dk nodes) queued %.3f synced %.3f Mp/s %.0f" % (n / 1e6, st.n_crit // 1000, queued, float(np.median(ms)), n / queued / 1e3))
    del st, outs
print("; ".join(out))
