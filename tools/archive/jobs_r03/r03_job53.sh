#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rv in 0 60000; do
  echo "== RK_PLAN_REV_MAX_GROUPS=$rv"
  RK_PLAN_REV_MAX_GROUPS=$rv timeout 600 python3 tools/shard_sim.py 4000000 2>&1 | grep -E "full|N=2|N=4"
  RK_PLAN_REV_MAX_GROUPS=$rv timeout 600 python3 tools/any_probe3.py 2>&1 | tail -1 | cut -c1-60
  RK_PLAN_REV_MAX_GROUPS=$rv timeout 600 python3 - <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, rakau_amd
from bench import plummer_numpy
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
for n in (1_200_000, 1_500_000, 1_800_000, 2_000_000, 2_200_000):
    m, x, y, z = plummer_numpy(n, "float32")
    st = rakau_amd.Octree(x, y, z, m).state()
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(40):
        st.acc_pot_device(0, mv, ptrs)
    st.set_timing(False); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        st.acc_pot_device(0, mv, ptrs)
    e1.record(); torch.cuda.synchronize()
    print("  %dk (%d nodes): %.4f ms per queued call" % (n // 1000, st.n_crit, e0.elapsed_time(e1) / 100))
PY
done
