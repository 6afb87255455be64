"""A fresh process: 60k-particle state, a few device-output calls, exit. Looped by a job script to catch a rare crash in
the first calls of a process (run with RK_BACKTRACE=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, oracle, rakau_amd
from helpers import state_from_oracle
n = 60000
m, x, y, z = oracle.plummer(n, np.float32)
ot = oracle.Tree(x, y, z, m)
st = state_from_oracle(ot)
st.set_perm(ot.codes_perms()["perm"])
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
for variant in (0, 2, 3, 4):
    st.set_variant(variant)
    for q in (0, 2):
        outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(rakau_amd.NRES[q])]
        for rep in range(3):
            st.acc_pot_device(q, mv, [o.data_ptr() for o in outs], eps2=1e-6)
        torch.cuda.synchronize()
print("ok")
