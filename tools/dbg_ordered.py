import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import oracle, rakau_amd
dev = torch.device("cuda", 0)
dtype = np.float32
m, x, y, z = oracle.plummer(30000, dtype)
for rep in range(12):
    q = 0
    st = rakau_amd.State.build(x, y, z, m)
    mv = rakau_amd.mac_value_of(0.75, "bh", dtype)
    perm = st.download("perm").astype(np.int64)
    tt = torch.float32
    ref = st.acc_pot(q, mv, eps2=1e-6)
    outs = [torch.full((st.nparts,), -7.0, dtype=tt, device=dev) for _ in ref]
    st.acc_pot_device(q, mv, [o.data_ptr() for o in outs], eps2=1e-6, ordered=True)
    torch.cuda.synchronize()
    third = st.acc_pot(q, mv, eps2=1e-6)
    fourth = st.acc_pot(q, mv, eps2=1e-6)
    o = [t.cpu().numpy()[perm] for t in outs]
    cr = st.crit_ranges()
    print(rep, "first!=third", [int((a != b).sum()) for a, b in zip(ref, third)], "ordered!=third", [int((a != b).sum()) for a, b in zip(o, third)],
          "third!=fourth", [int((a != b).sum()) for a, b in zip(fourth, third)])
    bad = np.nonzero(ref[0] != third[0])[0]
    if bad.size:
        g = np.unique(np.searchsorted(cr[:, 0], bad, side="right") - 1)
        print("   groups", g[:12], "n groups", g.size, "sizes", (cr[g[:12], 1] - cr[g[:12], 0]))
