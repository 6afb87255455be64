import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rakau_amd
from bench import plummer_numpy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
rakau_amd.State.build(x[:2000], y[:2000], z[:2000], m[:2000]).close()
for _ in range(3):
    t0 = time.perf_counter(); s = rakau_amd.State.build(x, y, z, m); dt = time.perf_counter() - t0
    print("State.build %d particles: %.2f ms, %d nodes, %d groups" % (n, dt * 1e3, s.tree_size, s.n_crit)); s.close()
