"""Run-to-run determinism probe: same call repeated, host path vs device path vs ordered path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import oracle, rakau_amd
dtype = np.float32
m, x, y, z = oracle.plummer(30000, dtype)
st = rakau_amd.State.build(x, y, z, m)
mv = rakau_amd.mac_value_of(0.75, "bh", dtype)
dev = torch.device("cuda", 0)
a = st.acc_pot(0, mv, eps2=1e-6)
b = st.acc_pot(0, mv, eps2=1e-6)
print("host vs host:", [int((u != v).sum()) for u, v in zip(a, b)])
outs = [torch.zeros(st.nparts, dtype=torch.float32, device=dev) for _ in range(3)]
for rep in range(3):
    st.acc_pot_device(0, mv, [o.data_ptr() for o in outs], eps2=1e-6)
    torch.cuda.synchronize()
    print("host vs device rep", rep, [int((u != o.cpu().numpy()).sum()) for u, o in zip(a, outs)])
perm = st.download("perm").astype(np.int64)
for rep in range(3):
    st.acc_pot_device(0, mv, [o.data_ptr() for o in outs], eps2=1e-6, ordered=True)
    torch.cuda.synchronize()
    print("host vs ordered rep", rep, [int((u != o.cpu().numpy()[perm]).sum()) for u, o in zip(a, outs)])
