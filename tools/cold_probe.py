#!/usr/bin/env python3
"""Where does the first call's time go? A fresh process, timestamps around every first step: library load, first HIP call,
state creation of a tiny tree (code objects, streams, pool), first traversal, then the 4M state and its first calls."""
import os, sys, time
t00 = time.perf_counter()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
T = {}
def lap(name, t0):
    T[name] = time.perf_counter() - t0
    print("%-44s %8.3f s" % (name, T[name]), flush=True)
t0 = time.perf_counter(); import rakau_amd; from rakau_amd import _capi; lap("import rakau_amd", t0)
t0 = time.perf_counter(); L = _capi.lib(); lap("dlopen librakau_amd.so", t0)
t0 = time.perf_counter(); n_dev = L.rk_device_count(); lap("rk_device_count (HIP runtime init)", t0)
from bench import plummer_numpy
m, x, y, z = plummer_numpy(20000, "float32")
t0 = time.perf_counter(); t = rakau_amd.Octree(x, y, z, m); lap("host tree, 20k particles", t0)
t0 = time.perf_counter(); st = t.state(); lap("first rk_state_create (20k)", t0)
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
t0 = time.perf_counter(); st.acc_pot(0, mv); lap("first rk_acc_pot (20k)", t0)
t0 = time.perf_counter(); st.acc_pot(0, mv); lap("second rk_acc_pot (20k)", t0)
t0 = time.perf_counter(); st.acc_pot(2, mv); lap("first accs_pots (20k, other kernels)", t0)
n = int(float(os.environ.get("N", "4e6")))
m, x, y, z = plummer_numpy(n, "float32")
t0 = time.perf_counter(); t4 = rakau_amd.Octree(x, y, z, m); lap("host tree, %d particles" % n, t0)
t0 = time.perf_counter(); s4 = t4.state(); lap("rk_state_create (%d)" % n, t0)
outs = [rakau_amd.pinned_empty(n, np.float32) for _ in range(3)]
for i in range(4):
    t0 = time.perf_counter(); s4.acc_pot(0, mv, out=outs); lap("rk_acc_pot #%d (%d, pinned outputs)" % (i + 1, n), t0)
print("total %.2f s" % (time.perf_counter() - t00))
