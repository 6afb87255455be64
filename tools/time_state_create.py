"""Phase timing of rk_state_create (host tree -> device state), cold and warm: RK_BUILD_TIMING=1 python tools/time_state_create.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rakau_amd
from bench import plummer_numpy
m, x, y, z = plummer_numpy(4_000_000, "float32")
t = rakau_amd.Octree(x, y, z, m)
nodes = t.nodes()
px, py, pz, pm = t.p_its_u()
for rep in range(3):
    t0 = time.perf_counter()
    st = rakau_amd.State(px, py, pz, pm, nodes, ncrit=128)
    print("rk_state_create #%d: %.1f ms" % (rep, (time.perf_counter() - t0) * 1e3), flush=True)
    st.close()
