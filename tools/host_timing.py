#!/usr/bin/env python3
"""Timeline of the chunked kernel / device-to-host pipeline of rk_acc_pot(): RK_HOST_TIMING=1 RK_HOST_CHUNKS=<n> python3 tools/host_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rakau_amd
from bench import plummer_numpy
n = 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
t = rakau_amd.Octree(x, y, z, m)
st = t.state()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
out = [np.zeros(n, dtype=np.float32) for _ in range(3)]
for i in range(5):
    t0 = time.perf_counter()
    st.acc_pot(0, mv, out=out)
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
