#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_device_build.py tests/test_gpu_quadtree.py tests/test_gpu_leapfrog.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -8
cd /tmp && export TMPDIR=/tmp
cat > /tmp/exact_probe.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch, rakau_amd, time
from bench import plummer_numpy
from rakau_amd import _capi
n = 4000000
m, x, y, z = plummer_numpy(n, "float32")
ts = [torch.as_tensor(v).cuda() for v in (x, y, z, m)]
torch.cuda.synchronize()
_capi.lib().rk_set_build_exact(1)
st = rakau_amd.State.build_device([t.data_ptr() for t in ts], n, np.float32)
for _ in range(3):
    t0 = time.perf_counter(); st.rebuild_device([t.data_ptr() for t in ts]); torch.cuda.synchronize(); print("exact rebuild ms", (time.perf_counter() - t0) * 1e3)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_exact -- python3 /tmp/exact_probe.py 2>&1 | grep "exact rebuild"
f=$(find /tmp/prof_exact -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r["Name"][:60], r["Calls"], "total ms %.3f" % (float(r["TotalDurationNs"]) / 1e6), "avg us %.1f" % (float(r["AverageNs"]) / 1e3), "max us %.1f" % (float(r["MaxNs"]) / 1e3))
PY
g=$(find /tmp/prof_exact -name "*kernel_trace.csv" | head -1)
python3 - "$g" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_exact_chains" in r["Kernel_Name"]]
rows = rows[-2:]
for r in rows:
    print(("wave " if "exact_wave" in r["Kernel_Name"] else "thread"), "%.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
