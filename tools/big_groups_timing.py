#!/usr/bin/env python3
"""Timing of critical nodes too large for one wavefront: uniform particles, ncrit far above 256 so that every node takes
the big-node path. RK_BIG_DFS=1 selects the scalar block-per-node walk, the default is the chunked list kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500000
ncrit = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rng = np.random.default_rng(3)
x, y, z = (rng.uniform(-0.5, 0.5, n).astype(np.float32) for _ in range(3))
m = rng.uniform(0.1, 1.0, n).astype(np.float32)
st = rakau_amd.State.build(x, y, z, m, box_size=1.0, ncrit=ncrit)
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ms = []
for _ in range(5):
    st.acc_pot_device(0, mv, [o.data_ptr() for o in outs])
    ms.append(st.last_kernel_ms())
c = st.count_interactions(mv)
cr = st.crit_ranges()
print("n=%d ncrit=%d critical nodes %d (largest %d) interactions/particle %.0f kernel ms %.2f -> %.2e interactions/s  [RK_BIG_DFS=%s] checksum %.9e"
      % (n, ncrit, len(cr), int((cr[:, 1] - cr[:, 0]).max()), (c["com"] + c["pp"] + c["self"]) / n, min(ms),
         (c["com"] + c["pp"] + c["self"]) / (min(ms) * 1e-3), os.environ.get("RK_BIG_DFS", "0"), float(outs[0].double().abs().sum())))
