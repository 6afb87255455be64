"""Traversal time of a 2-D (quadtree) problem: 2M particles in a disc, theta = 0.75, fp32."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rakau_amd
rng = np.random.default_rng(1)
n = 2_000_000
r = np.sqrt(rng.random(n)) ; ph = rng.uniform(0, 2 * np.pi, n)
x, y = (r * np.cos(ph)).astype(np.float32), (r * np.sin(ph)).astype(np.float32)
m = rng.uniform(0.5, 1.5, n).astype(np.float32)
st = rakau_amd.State.build(x, y, None, m)
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
import torch
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(2)]
ms = []
for _ in range(12):
    st.acc_pot_device(0, mv, [o.data_ptr() for o in outs])
    ms.append(st.last_kernel_ms())
c = st.count_interactions(mv)
print("2-D 2M: kernel %.3f ms, %.0f Mparticles/s, %.0f interactions/particle" % (np.median(ms[3:]), n / np.median(ms[3:]) / 1e3, (c["com"] + c["pp"] + c["self"]) / n))
