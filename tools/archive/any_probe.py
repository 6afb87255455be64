#!/usr/bin/env python3
"""Kernel ms (repeated calls: launch plan + graph) of 100k, 350k, 1M trees and of the 8 shards of the 4M tree under the
one-launch kernels (RK_ANY: 0 = class launches, 1 = k_pc_any, 2 = k_pc for R = 2 + k_list_any, 3 = k_list_any, unset =
automatic), with a hash of the result bits per case (must not depend on RK_ANY)."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy, shard_cuts
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
def timed(st, ptrs, b, e, reps=16):
    ms = []
    st.set_timing(True)
    for _ in range(reps):
        st.acc_pot_device(0, mv, ptrs, p_begin=b, p_end=e, offset_output=False)
        ms.append(st.last_kernel_ms())
    return float(np.median(ms[5:]))
def digest(outs, n):
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for o in outs:
        h.update(o[:n].cpu().numpy().tobytes())
    return h.hexdigest()[:10]
res = []
for n in (100_000, 350_000, 1_000_000):
    m, x, y, z = plummer_numpy(n, "float32")
    st = rakau_amd.Octree(x, y, z, m).state()
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(40):
        st.acc_pot_device(0, mv, ptrs)
    t = timed(st, ptrs, 0, n, 30)
    res.append("%dk(%d nodes) %.4f %s" % (n // 1000, st.n_crit, t, digest(outs, n)))
    del st, outs
n = 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
for _ in range(40):
    st.acc_pot_device(0, mv, ptrs)
cuts = shard_cuts(st.crit_ranges(), n, 8, st.group_work(mv))
sh, hs = [], hashlib.sha1()
for r in range(8):
    sh.append(timed(st, ptrs, cuts[r], cuts[r + 1]))
    hs.update(digest(outs, cuts[r + 1] - cuts[r]).encode())
res.append("8 shards max %.4f mean %.4f %s" % (max(sh), float(np.mean(sh)), hs.hexdigest()[:10]))
print("RK_ANY=%s | %s" % (os.environ.get("RK_ANY", "auto"), " | ".join(res)), flush=True)
