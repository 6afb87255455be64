#!/usr/bin/env python3
"""Producer / consumer kernel (variants 3, 4) against the list kernel (variant 2): bit-identity of variant 3, rounding-level
agreement and determinism of variant 4, kernel times. usage: pc_check.py [nparts...]"""
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
        for v in (2, 3, 4):
            st.set_variant(v)
            ms = []
            for _ in range(8):
                st.acc_pot_device(q, mv, ptrs, eps2=1e-6 if q else 0.0)
                ms.append(st.last_kernel_ms())
            torch.cuda.synchronize()
            res[(q, v)] = ([o.cpu().numpy().copy() for o in outs], float(np.median(ms[3:])))
        a, b, c = res[(q, 2)][0], res[(q, 3)][0], res[(q, 4)][0]
        same3 = all(np.array_equal(u, w) for u, w in zip(a, b))
        err4 = max(float(np.max(np.abs(u.astype(np.float64) - w) / (np.abs(u.astype(np.float64)).max()))) for u, w in zip(a, c))
        # determinism of variant 4
        st.set_variant(4)
        outs2 = [torch.zeros(n, dtype=tt, device="cuda") for _ in range(rakau_amd.NRES[q])]
        st.acc_pot_device(q, mv, [o.data_ptr() for o in outs2], eps2=1e-6 if q else 0.0)
        torch.cuda.synchronize()
        det4 = all(np.array_equal(u, o.cpu().numpy()) for u, o in zip(c, outs2))
        print("%s n=%d q=%d: v2 %.3f ms, v3 %.3f ms (bit-identical: %s), v4 %.3f ms (max |diff|/max|a| %.2e, deterministic: %s)"
              % (dtype, n, q, res[(q, 2)][1], res[(q, 3)][1], same3, res[(q, 4)][1], err4, det4), flush=True)
