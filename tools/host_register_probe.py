#!/usr/bin/env python3
"""rk_acc_pot() into pageable host arrays at 4M: the staging path against registering the caller's arrays per call
(RK_HOST_REGISTER=1). Checks the results against a device-output call, and times calls into fresh arrays too."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy
n = int(float(os.environ.get("N", "4e6")))
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
dev = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
st.acc_pot_device(0, mv, [d.data_ptr() for d in dev]); torch.cuda.synchronize()
ref = [d.cpu().numpy() for d in dev]
out = [np.zeros(n, np.float32) for _ in range(3)]
ts = []
for _ in range(25):
    t0 = time.perf_counter(); st.acc_pot(0, mv, out=out); ts.append((time.perf_counter() - t0) * 1e3)
same = all(np.array_equal(a, b) for a, b in zip(out, ref))
fresh = []
for _ in range(8):
    o2 = [np.zeros(n, np.float32) for _ in range(3)]   # new, touched pages at new addresses
    t0 = time.perf_counter(); st.acc_pot(0, mv, out=o2); fresh.append((time.perf_counter() - t0) * 1e3)
    same = same and all(np.array_equal(a, b) for a, b in zip(o2, ref))
# a sub-range into the middle of unaligned arrays
cr = st.crit_ranges(); b, e = int(cr[len(cr) // 3, 0]), int(cr[2 * len(cr) // 3, 0])
o3 = [np.zeros(n + 3, np.float32)[3:] for _ in range(3)]
st.acc_pot(0, mv, out=o3, p_begin=b, p_end=e)
same = same and all(np.array_equal(a[b:e], r[b:e]) and not a[:b].any() and not a[e:].any() for a, r in zip(o3, ref))
print("RK_HOST_REGISTER=%s: same arrays ms per call median %.3f min %.3f first %.3f; fresh arrays median %.3f; kernel %.3f; results identical to the device-output call: %s"
      % (os.environ.get("RK_HOST_REGISTER", "0"), float(np.median(ts[8:])), min(ts), ts[0], float(np.median(fresh)), st.last_kernel_ms(), same))
