"""What one time step costs a caller that goes through the ROCm seam with a tree it rebuilds on the host every step (the
reference's update_particles(): rocm_reset_state -> host rebuild -> rocm_init_state): rk_state_create from the host arrays, one
rk_acc_pot into pageable arrays, rk_state_destroy. The host tree itself is built once here (its cost is the caller's).
    python tools/seam_step_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rakau_amd
from bench import plummer_numpy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
t = rakau_amd.Octree(x, y, z, m)
p = t.p_its_u()
nodes = t.nodes()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
out = [np.zeros(n, dtype=np.float32) for _ in range(3)]
rows = []
for step in range(8):
    t0 = time.perf_counter()
    s = rakau_amd.State(p[0], p[1], p[2], p[3], nodes, ncrit=128)
    t1 = time.perf_counter()
    s.acc_pot(0, mv, out=out)
    t2 = time.perf_counter()
    s.close()
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print("n = %d: per step ms  create / acc_pot (pageable) / destroy" % n)
for r in rows:
    print("   %7.2f %7.2f %7.2f   total %7.2f" % (r + (sum(r),)))
