"""rk_set_common_eval(state, 1): the supergroups' common source lists evaluated once per supergroup by k_common (4 targets
per lane on full lanes) instead of by every member node; the members start from its per-particle sums. Not the default
(slower on MI355X) -- an alternative summation order of the same interaction set, checked here against
the CPU oracle and for the bit-identities the default mode guarantees: list kernel == producer / consumer kernel, union of
shards == full range, repeated calls, and a state switched back and forth."""
import numpy as np
import pytest

import oracle
from helpers import state_from_oracle, rel_err_vec, rel_err
from rakau_amd import mac_value_of

pytestmark = pytest.mark.gpu

TOL = {np.float32: 2e-5, np.float64: 1e-12}


def _err(got, ref, q):
    e = 0.0
    if q in (0, 2):
        e = max(e, rel_err_vec(got, ref).max())
    if q in (1, 2):
        e = max(e, rel_err(got[-1], ref[-1]).max())
    return e


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
def test_common_eval_matches_oracle_and_itself(dtype, mac):
    n = 20000
    m, x, y, z = oracle.plummer(n, dtype)
    ot = oracle.Tree(x, y, z, m, mac=mac)
    mv = mac_value_of(0.75, mac, dtype)
    eps2 = float(dtype(0.01) ** 2)
    for q in (0, 1, 2):
        ref = ot.acc_pot(q, 0.75, eps=0.01, nthreads=8)
        res = {}
        for var in (2, 3):
            st = state_from_oracle(ot)
            st.set_variant(var)
            st.set_common_eval(1)
            got = st.acc_pot(q, mv, eps2=eps2)
            assert _err(got, ref, q) <= TOL[dtype]
            for a, b in zip(got, st.acc_pot(q, mv, eps2=eps2)):
                assert np.array_equal(a, b)
            # Five Morton shards: their union is the full-range result bit for bit.
            cr = st.crit_ranges()
            cuts = [0] + [int(cr[len(cr) * k // 5, 0]) for k in range(1, 5)] + [n]
            parts = [st.acc_pot(q, mv, eps2=eps2, p_begin=cuts[k], p_end=cuts[k + 1], offset_output=False) for k in range(5)]
            for k, g in enumerate(got):
                assert np.array_equal(g, np.concatenate([p[k] for p in parts]))
            res[var] = got
        for a, b in zip(res[2], res[3]):
            assert np.array_equal(a, b)


def test_common_eval_switching_and_parameters():
    """One state switched between the two modes, with q, eps and G changing between calls (the cached pre-pass output is
    keyed on what it was made for): every result equals that of a fresh state in the same mode; G scales exactly."""
    dtype = np.float32
    m, x, y, z = oracle.plummer(30000, dtype)
    ot = oracle.Tree(x, y, z, m)
    mv = mac_value_of(0.75, "bh", dtype)
    st = state_from_oracle(ot)

    def fresh(mode, q, eps2, G):
        f = state_from_oracle(ot)
        f.set_common_eval(mode)
        return f.acc_pot(q, mv, eps2=eps2, G=G)

    seq = [(1, 0, 0.0, 1.0), (1, 0, 0.0, 1.0), (1, 2, 0.0, 1.0), (1, 2, 1e-4, 1.0), (0, 2, 1e-4, 1.0), (1, 2, 1e-4, 2.0), (1, 1, 1e-4, 2.0),
           (0, 0, 0.0, 1.0), (1, 0, 0.0, 1.0)]
    for mode, q, eps2, G in seq:
        st.set_common_eval(mode)
        got = st.acc_pot(q, mv, eps2=eps2, G=G)
        for a, b in zip(got, fresh(mode, q, eps2, G)):
            assert np.array_equal(a, b), (mode, q, eps2, G)
    a1 = fresh(1, 2, 1e-4, 1.0)
    a2 = fresh(1, 2, 1e-4, 2.0)
    for u, v in zip(a1, a2):
        assert np.array_equal(u * dtype(2), v)
    # The two modes differ by rounding only.
    b0, b1 = fresh(0, 0, 0.0, 1.0), fresh(1, 0, 0.0, 1.0)
    assert rel_err_vec(b0, b1).max() < 1e-5


def test_common_eval_big_groups():
    """Critical nodes too large for one wavefront (chunked BIG kernel) start from the sums as well."""
    rng = oracle.Rng(7)
    dtype = np.float32
    m, x, y, z = rng.uniform_particles(6000, 1.0, dtype)
    x[:1500], y[:1500], z[:1500] = 0.123, -0.2, 0.31
    for max_leaf_n, ncrit in ((16, 300), (700, 5000)):
        ot = oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=max_leaf_n, ncrit=ncrit)
        ref = ot.acc_pot(2, 0.6, eps=0.01, nthreads=8)
        st = state_from_oracle(ot)
        st.set_common_eval(1)
        got = st.acc_pot(2, mac_value_of(0.6, "bh", dtype), eps2=float(dtype(0.01) ** 2))
        assert _err(got, ref, 2) <= 1e-4
