import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, rakau_amd
from bench import plummer_numpy
n = int(float(sys.argv[1]))
variant = int(sys.argv[2])
m, x, y, z = plummer_numpy(n, "float32")
st = rakau_amd.Octree(x, y, z, m).state()
st.set_variant(variant)
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
for _ in range(6):
    st.acc_pot_device(0, mv, [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
print("n", n, "variant", variant, "n_crit", st.n_crit, "kernel ms", st.last_kernel_ms())
