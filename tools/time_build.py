import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rakau_amd
from bench import plummer_numpy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
rakau_amd.Octree(x[:2000], y[:2000], z[:2000], m[:2000], builder="device").close()
for b in ("device", "host", "device", "host"):
    t0 = time.perf_counter(); t = rakau_amd.Octree(x, y, z, m, builder=b); t1 = time.perf_counter()
    st = t.state(); t2 = time.perf_counter()
    print("%s: Octree ctor %.1f ms, state() %.1f ms" % (b, (t1 - t0) * 1e3, (t2 - t1) * 1e3)); t.close()
t0 = time.perf_counter(); s = rakau_amd.State.build(x, y, z, m); print("State.build only %.1f ms" % ((time.perf_counter() - t0) * 1e3))
