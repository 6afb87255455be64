"""rk_acc_pot() into PAGEABLE arrays at 4M (or argv[1]) particles: ms per blocking call (median of the last calls after the clocks have
settled), kernel ms, and the bits against the device-output call. Run under RK_HOST_SPLIT=<fraction> (0 = one part)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ["RK_SUPER_CACHE"] = "0"
m, x, y, z = plummer_numpy(n, "float32")
t = rakau_amd.Octree(x, y, z, m)
st = t.state()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
nres = rakau_amd.NRES[q]
out = [np.zeros(n, dtype=np.float32) for _ in range(nres)]
ts, ks = [], []
while len(ts) < 12 or sum(ts[2:]) < 0.15:
    t0 = time.perf_counter()
    st.acc_pot(q, mv, out=out)
    ts.append(time.perf_counter() - t0)
    ks.append(st.last_kernel_ms())
d = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(nres)]
st.acc_pot_device(q, mv, [v.data_ptr() for v in d])
torch.cuda.synchronize()
same = all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(out, d))
print("RK_HOST_SPLIT=%s n=%d q=%d: %.4f ms per call (kernels %.4f), same bits as the device-output call: %s"
      % (os.environ.get("RK_HOST_SPLIT", "default"), n, q, float(np.median(ts[-10:])) * 1e3, float(np.median(ks[-10:])), same))
