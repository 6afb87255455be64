import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
from helpers import state_from_oracle
from rakau_amd import mac_value_of
n = int(sys.argv[1]); dtype = np.float64 if sys.argv[2]=="d" else np.float32; seed=int(sys.argv[3])
m, x, y, z = oracle.plummer(n, dtype, seed=seed)
ot = oracle.Tree(x, y, z, m)
st = state_from_oracle(ot); st.set_variant(2)
print("groups", st.n_crit, "max", st.max_group, flush=True)
got = st.acc_pot(0, mac_value_of(0.75, "bh", dtype))
ref = ot.acc_pot(0, 0.75, nthreads=8)
print("ok", np.abs(got[0]-ref[0]).max(), flush=True)
