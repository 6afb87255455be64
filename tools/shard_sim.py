#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: time every rank's shard of the 4M bench problem by itself and report
max-over-shards (what an N-GPU run would take per step, excluding launch skew) for particle-balanced and
work-balanced cuts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy, shard_cuts

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
m, x, y, z = plummer_numpy(n, "float32")
t = rakau_amd.Octree(x, y, z, m)
st = t.state()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
crit = st.crit_ranges()
work = st.group_work(mv)
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]

def time_range(b, e, reps=12):
    ms = []
    for _ in range(reps):
        st.acc_pot_device(0, mv, ptrs, p_begin=b, p_end=e, offset_output=False)
        ms.append(st.last_kernel_ms())
    return float(np.median(ms[2:]))

# Leave the idle power state first (a 4M call needs ~15 calls after idling before its duration settles).
for _ in range(40):
    st.acc_pot_device(0, mv, ptrs, p_begin=0, p_end=n, offset_output=False)
torch.cuda.synchronize()

for variant in variants:
    st.set_variant(variant)
    full = time_range(0, n, reps=24)
    print("variant %d full range: %.3f ms" % (variant, full))
    for world in (2, 4, 8):
        for name, w in (("work", work),) if len(variants) > 1 else (("particles", None), ("work", work)):
            cuts = shard_cuts(crit, n, world, w)
            ts = [time_range(cuts[r], cuts[r + 1]) for r in range(world)]
            print("variant %d N=%d %-9s per-shard ms %s  max %.3f  ideal %.3f  efficiency %.2f" % (
                variant, world, name, " ".join("%.3f" % v for v in ts), max(ts), full / world, full / world / max(ts)))
