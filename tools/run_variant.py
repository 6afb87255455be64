#!/usr/bin/env python3
"""run_variant.py n variant [reps] [dtype]: repeated accs_u() calls of one kernel variant on a Plummer sphere (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy

n = int(float(sys.argv[1])); variant = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dtype = sys.argv[4] if len(sys.argv) > 4 else "float32"
theta = float(os.environ.get("THETA", "0.75"))
m, x, y, z = plummer_numpy(n, dtype)
t = rakau_amd.Octree(x, y, z, m)
st = t.state()
st.set_variant(variant)
mv = rakau_amd.mac_value_of(theta, "bh", np.dtype(dtype).type)
outs = [torch.zeros(n, dtype=getattr(torch, dtype), device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
ms = []
for _ in range(reps):
    st.acc_pot_device(0, mv, ptrs)
    ms.append(st.last_kernel_ms())
torch.cuda.synchronize()
print("n=%d variant %d kernel ms: median %.4f min %.4f" % (n, variant, float(np.median(ms[reps // 3:])), min(ms)), flush=True)
