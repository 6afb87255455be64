"""Stress of the capture -> instantiate -> launch path of repeated calls: every second call has a new key (other range /
Q / outputs), so the state destroys its graph, captures and launches a new one. Run with RK_BACKTRACE=1. With RK_PLAN=0
RK_GRAPH_FORKED=1 (graphs with parallel branches) the runtime dies inside hipGraphLaunch within 500 iterations.
    python tools/stress_graph_capture.py [iterations] [stream: 0 = legacy default stream, 1 = a torch stream]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, oracle, rakau_amd
from helpers import state_from_oracle
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
use_stream = len(sys.argv) > 2 and sys.argv[2] == "1"
n = 60000
m, x, y, z = oracle.plummer(n, np.float32)
ot = oracle.Tree(x, y, z, m)
st = state_from_oracle(ot)
cr = st.crit_ranges()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(4)]
ts = torch.cuda.Stream() if use_stream else None
sp = ts.cuda_stream if ts is not None else None
TIMING = os.environ.get('STRESS_TIMING', '1') == '1'
st.set_timing(TIMING)
t0 = time.time()
for it in range(iters):
    q = (0, 2)[it & 1]
    b = int(cr[(it * 7) % (len(cr) // 2), 0])
    e = int(cr[len(cr) // 2 + (it * 13) % (len(cr) // 2), 0])
    for rep in range(2 + (it % 3 == 0)):
        st.acc_pot_device(q, mv, [o.data_ptr() for o in outs[:rakau_amd.NRES[q]]], eps2=1e-6, p_begin=b, p_end=e, stream=sp)
    if it % 4 == 0:
        torch.cuda.synchronize()
    if it % 500 == 0:
        print('iteration', it, flush=True)
torch.cuda.synchronize()
print("graph stress ok: %d iterations in %.1f s" % (iters, time.time() - t0))
