#!/usr/bin/env python3
"""Per-wave timeline of the list kernel (diagnostic build -DRK_TRACE, tools/build_variant_full.sh trace -DRK_TRACE):
RAKAU_AMD_LIB=rakau_amd/lib_trace/librakau_amd.so python3 tools/trace_waves.py <out.npz> [nparts] [p_begin_frac p_end_frac]
Saves per critical node: start/end (10 ns ticks), hw id, xcc id, size, R, and the census work (rk_group_work)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
fr = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.0, 1.0)
tf = out + ".raw"
os.environ["RK_TRACE_FILE"] = tf
import torch
import rakau_amd
from bench import plummer_numpy, shard_cuts
m, x, y, z = plummer_numpy(n, "float32")
t = rakau_amd.Octree(x, y, z, m)
st = t.state()
if os.environ.get("VARIANT"):
    st.set_variant(int(os.environ["VARIANT"]))
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
work = st.group_work(mv)
cr = st.crit_ranges()
cum = np.cumsum(work.astype(np.float64))
def cut(f):
    if f <= 0: return 0
    if f >= 1: return n
    i = int(np.searchsorted(cum, cum[-1] * f)) + 1
    return int(cr[i, 0]) if i < len(cr) else n
pb, pe = cut(fr[0]), cut(fr[1])
outs = [torch.zeros(pe - pb, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
ms = []
for _ in range(6):
    st.acc_pot_device(0, mv, ptrs, p_begin=pb, p_end=pe, offset_output=False)
    ms.append(st.last_kernel_ms())
raw = np.fromfile(tf, dtype=np.uint64).reshape(-1, 4)
os.remove(tf)
new = (raw[:, 3] >> np.uint64(56)) != 0  # k_dense records {R, T, shader cycles}; the fused kernels {R, T}
Tn = np.where(new, (raw[:, 3] >> np.uint64(32)) & np.uint64(0xffffff), raw[:, 3] & np.uint64(0xffffffff)).astype(np.uint32)
Rn = np.where(new, raw[:, 3] >> np.uint64(56), raw[:, 3] >> np.uint64(32)).astype(np.uint8)
cyc = np.where(new, raw[:, 3] & np.uint64(0xffffffff), 0).astype(np.uint64)
np.savez_compressed(out, t0=raw[:, 0], t1=raw[:, 1], hw=(raw[:, 2] & np.uint64(0xffffffff)).astype(np.uint32),
                    xcc=(raw[:, 2] >> np.uint64(32)).astype(np.uint8), T=Tn, R=Rn, cyc=cyc, work=work, crit=cr,
                    kernel_ms=np.array(ms), range=np.array([pb, pe]))
print("kernel ms", ms, "range", pb, pe)
