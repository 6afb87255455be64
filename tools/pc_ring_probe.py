#!/usr/bin/env python3
"""Kernel ms of repeated full-range calls served by k_pc_any (RK_ANY=1 forced by the caller's environment) at a few sizes,
with a hash of the result bits: A/B of the producer / consumer hand-off (RK_PC_NB tile buffers) across library builds."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy
sizes = [int(float(a)) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [30_000, 100_000, 150_000, 350_000]
dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
mv = rakau_amd.mac_value_of(0.75, "bh", np.dtype(dtype).type)
res = []
for n in sizes:
    m, x, y, z = plummer_numpy(n, dtype)
    st = rakau_amd.Octree(x, y, z, m).state()
    outs = [torch.zeros(n, dtype=getattr(torch, dtype), device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(30):
        st.acc_pot_device(0, mv, ptrs)
    st.set_timing(True)
    ms = []
    for _ in range(40):
        st.acc_pot_device(0, mv, ptrs)
        ms.append(st.last_kernel_ms())
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for o in outs:
        h.update(o.cpu().numpy().tobytes())
    res.append("%dk(%d) %.4f min %.4f %s" % (n // 1000, st.n_crit, float(np.median(ms[5:])), min(ms), h.hexdigest()[:8]))
    del st, outs
print("%s RK_ANY=%s %s | %s" % (os.path.basename(os.path.dirname(os.environ.get("RAKAU_AMD_LIB", "lib/x"))), os.environ.get("RK_ANY", "auto"), dtype, " | ".join(res)), flush=True)
