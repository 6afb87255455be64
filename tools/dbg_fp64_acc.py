"""fp64 accuracy of the list kernel against the oracle and the direct sum (for RK_RSQ64_STEPS experiments)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle, rakau_amd
from tests.helpers import state_from_oracle, rel_err_vec
m, x, y, z = oracle.plummer(200000, np.float64)
ot = oracle.Tree(x, y, z, m)
st = state_from_oracle(ot)
mv = rakau_amd.mac_value_of(0.5, "bh", np.float64)
got = st.acc_pot(2, mv)
ref = ot.acc_pot(2, 0.5, nthreads=16)
e = rel_err_vec(got, ref)
print("vs oracle: median %.3e  99.9%% %.3e  max %.3e" % (np.median(e), np.percentile(e, 99.9), e.max()))
