#!/usr/bin/env python3
"""Producer / consumer kernel (variant 3) against the list kernel (variant 2): bit-identity and kernel times.
usage: pc_check.py [nparts...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy
sizes = [int(v) for v in sys.argv[1:]] or [100_000, 1_000_000, 4_000_000]
for dtype, n, theta in [("float32", s, 0.75) for s in sizes] + [("float64", 300_000, 0.5)]:
    m, x, y, z = plummer_numpy(n, dtype)
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(theta, "bh", dtype)
    tt = torch.float32 if dtype == "float32" else torch.float64
    res = {}
    for q in (0, 2):
        outs = [torch.zeros(n, dtype=tt, device="cuda") for _ in range(rakau_amd.NRES[q])]
        ptrs = [o.data_ptr() for o in outs]
        for v in (2, 3):
            st.set_variant(v)
            ms = []
            for _ in range(8):
                st.acc_pot_device(q, mv, ptrs, eps2=1e-6 if q else 0.0)
                ms.append(st.last_kernel_ms())
            torch.cuda.synchronize()
            res[(q, v)] = ([o.cpu().numpy().copy() for o in outs], float(np.median(ms[3:])))
        a, b = res[(q, 2)][0], res[(q, 3)][0]
        same3 = all(np.array_equal(u, w) for u, w in zip(a, b))
        print("%s n=%d q=%d: list kernel %.3f ms, producer/consumer %.3f ms (bit-identical: %s)"
              % (dtype, n, q, res[(q, 2)][1], res[(q, 3)][1], same3), flush=True)
