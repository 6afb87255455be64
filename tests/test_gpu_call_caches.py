"""The per-state caches of the launch path -- captured hipGraph of a repeated call, launch plan of a repeated small call,
supergroup pre-pass output reused while tree and MAC value stay the same -- must never change a result: sequences of calls
that alternate MAC values, ranges, Q and outputs on ONE state are compared, bit for bit, with the same calls made once
each on fresh replicas (rk_state_clone), which have no history."""
import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import state_from_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [6000, 60000])
def test_call_sequences_with_history_equal_fresh_calls(n):
    import torch
    dev = torch.device("cuda", 0)
    m, x, y, z = oracle.plummer(n, np.float32)
    ot = oracle.Tree(x, y, z, m)
    st = state_from_oracle(ot)
    st.set_perm(ot.codes_perms()["perm"])
    cr = st.crit_ranges()
    ng = len(cr)
    cuts = [0, int(cr[ng // 3, 0]), int(cr[2 * ng // 3, 0]), n]
    mv = {t: rakau_amd.mac_value_of(t, "bh", np.float32) for t in (0.75, 0.5)}
    bufs = {k: [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(4)] for k in ("a", "b")}

    def call(state, q, theta, rng, buf, ordered=False, eps2=0.0):
        outs = bufs[buf][:rakau_amd.NRES[q]]
        for o in outs:
            o.zero_()
        state.acc_pot_device(q, mv[theta], [o.data_ptr() for o in outs], eps2=eps2, p_begin=rng[0], p_end=rng[1],
                             offset_output=True, ordered=ordered)
        torch.cuda.synchronize()
        return [o.cpu().numpy().copy() for o in outs]

    full, r1, r2 = (0, n), (cuts[0], cuts[1]), (cuts[1], cuts[3])
    seq = [(0, 0.75, full, "a"), (0, 0.75, full, "a"), (0, 0.75, full, "a"),      # direct, capture, replay
           (0, 0.5, full, "a"), (0, 0.75, full, "a"), (0, 0.75, full, "a"),        # other MAC value in between
           (0, 0.75, r1, "a"), (0, 0.75, r1, "a"), (0, 0.75, r1, "a"),             # sub-range: plan + graph
           (0, 0.75, r2, "a"), (0, 0.75, r2, "a"), (0, 0.75, r1, "a"),             # another range rebuilds the plan
           (1, 0.75, r1, "a"), (1, 0.75, r1, "a"), (2, 0.75, r1, "b"), (2, 0.75, r1, "b"),  # Q and outputs change
           (0, 0.5, r2, "a"), (0, 0.5, r2, "a"), (0, 0.75, full, "a"), (0, 0.75, full, "b"), (0, 0.75, full, "b")]
    for i, (q, theta, rng, buf) in enumerate(seq):
        for ordered in ((False, True) if i % 5 == 0 else (False,)):
            got = call(st, q, theta, rng, buf, ordered=ordered, eps2=1e-6 if q else 0.0)
            fresh = st.clone(0)
            ref = call(fresh, q, theta, rng, "b" if buf == "a" else "a", ordered=ordered, eps2=1e-6 if q else 0.0)
            fresh.close()
            for g, r in zip(got, ref):
                assert np.array_equal(g, r), (i, q, theta, rng, ordered)
