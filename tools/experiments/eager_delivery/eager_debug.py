import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, oracle, rakau_amd
from helpers import state_from_oracle
n = 400000
m, x, y, z = oracle.plummer(n, np.float32)
ot = oracle.Tree(x, y, z, m)
st = state_from_oracle(ot)
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
cr = st.crit_ranges(); ng = len(cr)
for (b, e) in ((0, n), (int(cr[ng // 7, 0]), int(cr[(6 * ng) // 7, 0])), (0, n), (int(cr[ng // 7, 0]), int(cr[(6 * ng) // 7, 0]))):
    d = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    st.acc_pot_device(0, mv, [v.data_ptr() for v in d], eps2=1e-6, p_begin=b, p_end=e)
    torch.cuda.synchronize()
    out = st.acc_pot(0, mv, eps2=1e-6, p_begin=b, p_end=e, out=[np.zeros(n, dtype=np.float32) for _ in range(3)])
    for k in range(3):
        r = d[k].cpu().numpy()[b:e]; o = out[k][b:e]
        bad = np.nonzero(r.view(np.uint32) != o.view(np.uint32))[0]
        print("range", b, e, "array", k, "mismatches", len(bad), "first", bad[:8], "vals", o[bad[:4]], r[bad[:4]], "as hex", [hex(v) for v in o.view(np.uint32)[bad[:4]]], flush=True)
print("crit nodes:", ng)
import bisect
starts = [int(v) for v in cr[:, 0]]
for p in (27104, 27135, 58666 + 20896, 58666 + 20927, 58666 + 256992, 18528, 19824 + 58666):
    g = bisect.bisect_right(starts, p) - 1
    print("particle", p, "-> node", g, "range", cr[g], "size", int(cr[g, 1] - cr[g, 0]), "offset in node", p - int(cr[g, 0]), "prev size", int(cr[g - 1, 1] - cr[g - 1, 0]), "next size", int(cr[g + 1, 1] - cr[g + 1, 0]))
