#!/usr/bin/env python3
"""Where do the one-launch kernels stop paying? Kernel ms of 200k / 2M trees and of the 2 and 4 shards of the 4M tree."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy, shard_cuts
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
def timed(st, ptrs, b, e, reps=16):
    ms = []
    for _ in range(reps):
        st.acc_pot_device(0, mv, ptrs, p_begin=b, p_end=e, offset_output=False)
        ms.append(st.last_kernel_ms())
    return float(np.median(ms[5:]))
res = []
for n in (150_000, 200_000, 250_000, 2_000_000):
    m, x, y, z = plummer_numpy(n, "float32")
    st = rakau_amd.Octree(x, y, z, m).state()
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(40):
        st.acc_pot_device(0, mv, ptrs)
    res.append("%dk(%d) %.4f" % (n // 1000, st.n_crit, timed(st, ptrs, 0, n, 30)))
    del st, outs
n = 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
for _ in range(40):
    st.acc_pot_device(0, mv, ptrs)
w = st.group_work(mv)
for k in (2, 4):
    cuts = shard_cuts(st.crit_ranges(), n, k, w)
    sh = [timed(st, ptrs, cuts[r], cuts[r + 1]) for r in range(k)]
    res.append("%d shards max %.4f" % (k, max(sh)))
print("ANY=%s PCALL=%s PLANMAX=%s | %s" % (os.environ.get("RK_ANY", "auto"), os.environ.get("RK_PC_ALL_BELOW", "5000"), os.environ.get("RK_PLAN_MAX_GROUPS", "30000"), " | ".join(res)), flush=True)
