#!/usr/bin/env python3
"""Kernel time (HIP events, median of repeated calls) of accs_u() for a list of Plummer sizes: size_sweep.py [n,n,...] [dtype]."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy

sizes = [int(float(v)) for v in (sys.argv[1] if len(sys.argv) > 1 else "1e5,3.5e5,1e6,2e6,4e6").split(",")]
dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
for n in sizes:
    m, x, y, z = plummer_numpy(n, dtype)
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.dtype(dtype).type)
    outs = [torch.zeros(n, dtype=getattr(torch, dtype), device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    ms = []
    for _ in range(24):
        st.acc_pot_device(0, mv, ptrs)
        ms.append(st.last_kernel_ms())
    print("n=%d %s kernel ms: median %.4f min %.4f" % (n, dtype, float(np.median(ms[4:])), min(ms[4:])), flush=True)
    del st, t, outs
