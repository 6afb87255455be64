import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle, rakau_amd
mode = sys.argv[1]
rng = oracle.Rng(2)
m, x, y, z = rng.uniform_particles(10000, 1.0, np.float32)
if mode == "host_tree":
    t = rakau_amd.Octree(x, y, z, m, box_size=4.0)
elif mode == "host_tree_accs":
    t = rakau_amd.Octree(x, y, z, m, box_size=4.0); t.accs_o(0.5)
elif mode == "dev_tree":
    t = rakau_amd.Octree(x, y, z, m, box_size=4.0, builder="device")
elif mode == "dev_tree_accs":
    t = rakau_amd.Octree(x, y, z, m, box_size=4.0, builder="device"); t.accs_o(0.5)
elif mode == "state":
    t = rakau_amd.State.build(x, y, z, m)
elif mode == "exact":
    t = rakau_amd.Octree(x, y, z, m, box_size=4.0); t.exact_acc_o(3)
import torch
print(mode, "device_count", torch.cuda.device_count())
try:
    print(mode, torch.zeros(3, device="cuda").sum().item(), "OK")
except Exception as e:
    print(mode, "FAILED:", e)
