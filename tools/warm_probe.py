import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, rakau_amd
from bench import plummer_numpy
n = 4000000
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
stream = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
ts = []
for i in range(40):
    t0 = time.perf_counter()
    st.acc_pot_device(0, mv, ptrs, stream=stream)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("per-call wall ms (synchronised):", " ".join("%.2f" % t for t in ts))
# unsynchronised batches of 5
for rep in range(8):
    t0 = time.perf_counter()
    for _ in range(5):
        st.acc_pot_device(0, mv, ptrs, stream=stream)
    torch.cuda.synchronize()
    print("batch of 5: %.4f ms per call" % ((time.perf_counter() - t0) / 5 * 1e3))
