"""Stress of the executable-graph cache of repeated calls (rk_state.hip: gcache): a caller that keeps alternating among NSIG
recurring launch signatures (ranges x Q), `changes` key changes in all. With NSIG <= RK_GRAPH_CACHE (8) every signature is
captured once and replayed ever after; with NSIG > 8 the least recently used executables are evicted -- linear ones destroyed,
forked ones parked and re-targeted by hipGraphExecUpdate for the next capture -- and the number of forked executables alive stays
below RK_GRAPH_FORKED_MAX. Every result is compared with the first result of its signature (bit for bit).
    python tools/stress_graph_recurring.py [changes] [nsig] [calls per visit]
Run with RK_BACKTRACE=1. RK_PLAN=0 RK_ANY_FIRST=0 makes every sequence a forked one (class kernels on side streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, oracle, rakau_amd
from helpers import state_from_oracle

changes = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 8
per_visit = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 60000
m, x, y, z = oracle.plummer(n, np.float32)
st = state_from_oracle(oracle.Tree(x, y, z, m))
st.set_timing(False)
cr = st.crit_ranges()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
sigs = []
for i in range(nsig):
    b = int(cr[(i * 5) % (len(cr) // 3), 0])
    e = int(cr[len(cr) // 2 + (i * 11) % (len(cr) // 3), 0])
    q = (0, 2, 1)[i % 3]
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(rakau_amd.NRES[q])]
    sigs.append((q, b, e, outs))
first = {}
t0 = time.time()
rng = np.random.default_rng(3)
for it in range(changes):
    i = int(rng.integers(nsig)) if it >= 2 * nsig else it % nsig
    q, b, e, outs = sigs[i]
    for _ in range(per_visit):
        st.acc_pot_device(q, mv, [o.data_ptr() for o in outs], eps2=1e-6, p_begin=b, p_end=e)
    if it % 16 == 0 or i not in first:
        torch.cuda.synchronize()
        got = np.stack([o[b:e].cpu().numpy() for o in outs])
        if i not in first:
            first[i] = got
        elif not np.array_equal(first[i], got):
            print("MISMATCH at change %d, signature %d" % (it, i))
            sys.exit(1)
torch.cuda.synchronize()
gs = st.graph_stats()
print("graph cache stress ok: %d changes over %d signatures in %.1f s; %s" % (changes, nsig, time.time() - t0, gs))
calls = changes * per_visit
if nsig <= 8:
    # Every signature is launched directly once (twice where the pre-pass ran in one of the calls and was reused in the other:
    # that is part of the signature), captured once, replayed ever after.
    assert gs["direct"] <= 2 * nsig and gs["captures"] <= 2 * nsig and gs["replays"] >= calls - 4 * nsig, gs
else:
    assert gs["cached"] <= 8 and gs["forked_alive"] <= 64, gs
