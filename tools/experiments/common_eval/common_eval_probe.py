"""Probe of the pre-pass that evaluates the supergroups' common lists (rk_set_common_eval): parity against the CPU checker,
list kernel == producer / consumer kernel, shard union == full range, repeated calls, big groups -- with the mode forced on
small trees. Test infrastructure (uses the oracle)."""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from helpers import state_from_oracle, rel_err_vec, rel_err
from rakau_amd import mac_value_of


def main():
    bad = 0
    for dtype, tol in ((np.float32, 5e-5), (np.float64, 1e-12)):
        for n in (3000, 20000, 60000):
            m, x, y, z = oracle.plummer(n, dtype)
            for mac in ("bh", "bh_geom"):
                ot = oracle.Tree(x, y, z, m, mac=mac)
                mv = mac_value_of(0.75, mac, dtype)
                for q in (0, 1, 2):
                    ref = ot.acc_pot(q, 0.75, eps=0.01, nthreads=8)
                    eps2 = float(dtype(0.01) ** 2)
                    res = {}
                    for var in (2, 3):
                        for mode in (0, 1):
                            st = state_from_oracle(ot)
                            st.set_variant(var)
                            st.set_common_eval(mode)
                            got = st.acc_pot(q, mv, eps2=eps2)
                            again = st.acc_pot(q, mv, eps2=eps2)
                            res[(var, mode)] = got
                            e = 0.0
                            if q in (0, 2):
                                e = max(e, rel_err_vec(got, ref).max())
                            if q in (1, 2):
                                e = max(e, rel_err(got[-1], ref[-1]).max())
                            ok = e <= tol and all(np.array_equal(a, b) for a, b in zip(got, again))
                            if mode == 1:
                                # shards: union equals the full range bit for bit
                                cr = st.crit_ranges()
                                cuts = [0] + [int(cr[len(cr) * k // 5, 0]) for k in range(1, 5)] + [n]
                                parts = [st.acc_pot(q, mv, eps2=eps2, p_begin=cuts[k], p_end=cuts[k + 1], offset_output=False)
                                         for k in range(5) if cuts[k + 1] > cuts[k]]
                                for k, g in enumerate(got):
                                    ok = ok and np.array_equal(g, np.concatenate([p[k] for p in parts]))
                            if not ok:
                                bad += 1
                            print("%s n=%d %s q=%d variant %d common_eval %d: max err %.3g %s" % (dtype.__name__, n, mac, q, var, mode, e,
                                                                                           "ok" if ok else "FAIL"), flush=True)
                    for mode in (0, 1):
                        same = all(np.array_equal(a, b) for a, b in zip(res[(2, mode)], res[(3, mode)]))
                        if not same:
                            bad += 1
                            print("  list != pc in mode %d: FAIL" % mode)
    # big groups (chunked BIG kernel) in eval mode
    rng = oracle.Rng(7)
    m, x, y, z = rng.uniform_particles(6000, 1.0, np.float32)
    x[:1500], y[:1500], z[:1500] = 0.123, -0.2, 0.31
    for max_leaf_n, ncrit in ((16, 200), (16, 600), (700, 5000)):
        ot = oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=max_leaf_n, ncrit=ncrit)
        ref = ot.acc_pot(2, 0.6, eps=0.01, nthreads=8)
        st = state_from_oracle(ot)
        st.set_common_eval(1)
        got = st.acc_pot(2, mac_value_of(0.6, "bh", np.float32), eps2=float(np.float32(0.01) ** 2))
        e = max(rel_err_vec(got, ref).max(), rel_err(got[-1], ref[-1]).max())
        ok = e <= 1e-4
        bad += not ok
        print("big groups max_leaf_n=%d ncrit=%d: max err %.3g %s" % (max_leaf_n, ncrit, e, "ok" if ok else "FAIL"))
    print("common_eval probe: %d failure(s)" % bad)
    return bad != 0


if __name__ == "__main__":
    sys.exit(main())
