#!/usr/bin/env python3
"""Kernel ms of the full 4M range, of its eight 0.5M shards (max), of 1M and 100k trees under the launch order given by
RK_CLASS_ORDER (and the other launch-path knobs of the environment)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy, shard_cuts
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
def timed(st, ptrs, b, e, reps=14):
    ms = []
    for _ in range(reps):
        st.acc_pot_device(0, mv, ptrs, p_begin=b, p_end=e, offset_output=False)
        ms.append(st.last_kernel_ms())
    return float(np.median(ms[4:]))
res = []
n = 4_000_000
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
for _ in range(40):
    st.acc_pot_device(0, mv, ptrs)
torch.cuda.synchronize()
full = timed(st, ptrs, 0, n, 24)
cuts = shard_cuts(st.crit_ranges(), n, 8, st.group_work(mv))
sh = [timed(st, ptrs, cuts[r], cuts[r + 1]) for r in range(8)]
cuts2 = shard_cuts(st.crit_ranges(), n, 2, st.group_work(mv))
sh2 = [timed(st, ptrs, cuts2[r], cuts2[r + 1]) for r in range(2)]
res.append("4M %.3f | 2 shards max %.3f | 8 shards max %.3f mean %.3f" % (full, max(sh2), max(sh), float(np.mean(sh))))
del st, outs
for n in (1_000_000, 100_000):
    m, x, y, z = plummer_numpy(n, "float32")
    st = rakau_amd.Octree(x, y, z, m).state()
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(30):
        st.acc_pot_device(0, mv, ptrs)
    res.append("%dk %.4f" % (n // 1000, timed(st, ptrs, 0, n, 30)))
    del st, outs
print("order=%s %s" % (os.environ.get("RK_CLASS_ORDER", "default"), " | ".join(res)), flush=True)
