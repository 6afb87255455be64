"""accs_u() against accs_o() through the C++ header (rakau_amd::tree via the Python harness), ms per call at n particles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rakau_amd
from bench import plummer_numpy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
t = rakau_amd.Octree(x, y, z, m)
for name, fn in (("accs_u", t.accs_u), ("accs_o", t.accs_o)):
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); r = fn(0.75); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s n=%d: ms per call %s" % (name, n, " ".join("%.2f" % v for v in ts)))
u = t.accs_u(0.75); o = t.accs_o(0.75); perm = t.perm()
print("accs_o[perm] == accs_u:", all(np.array_equal(a[perm], b) for a, b in zip(o, u)))
