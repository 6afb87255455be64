"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on identical trees, every kernel variant (1 = depth-first
wave kernel, 2 = list kernel, 3 = producer / consumer kernel, 4 = split traversal (lists in HBM); 0 = automatic mixes 2
and 3 by launch size)."""
import numpy as np
import pytest

import oracle
from helpers import state_from_oracle, rel_err_vec, rel_err
from rakau_amd import mac_value_of

pytestmark = pytest.mark.gpu

# The HIP path evaluates the same interaction list as the oracle (critical-node groups); differences are
# rounding only. The reference's own bounds are fp32 |da|/|a| <= 2e-3 (test/ordering_acc.cpp:96) and
# fp64 <= 2e-11 (test/ordering_acc.cpp:94); identical lists deliver far better, so a tighter regression
# bound is asserted (~1e-7 median, < 1e-5 max in fp32).
TIGHT = {np.float32: 2e-5, np.float64: 1e-12}
# 0 = the automatic variant (what every caller gets: list kernel / producer-consumer kernel / one-launch forms chosen per call),
# 1 = scalar walk (cross-check library), 2 = list kernel, 3 = producer / consumer kernel, 4 = split traversal (cross-check library).
VARIANTS = [0, 1, 2, 3, 4]


def check(got, ref, q, dtype, tol=None, ndim=3):
    tol = TIGHT[dtype] if tol is None else tol
    for g in got:
        assert np.all(np.isfinite(g))
    if q in (0, 2):
        e = rel_err_vec(got, ref, ndim=ndim)
        assert e.max() <= tol, ("acc", e.max(), int(e.argmax()))
    if q in (1, 2):
        e = rel_err(got[-1], ref[-1])
        assert e.max() <= tol, ("pot", e.max(), int(e.argmax()))


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
@pytest.mark.parametrize("q", [0, 1, 2])
def test_plummer_20k(dtype, mac, q, variant):
    m, x, y, z = oracle.plummer(20000, dtype)
    theta = 0.75
    ot = oracle.Tree(x, y, z, m, mac=mac)
    ref = ot.acc_pot(q, theta, nthreads=8)
    st = state_from_oracle(ot)
    st.set_variant(variant)
    got = st.acc_pot(q, mac_value_of(theta, mac, dtype))
    check(got, ref, q, dtype)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_uniform_matrix(dtype, variant):
    """Tree-parameter matrix of the reference's accuracy tests (test/accuracy_acc.cpp:55-58): sizes x
    max_leaf_n x ncrit, at an opening angle that opens everything (0.001) and at 0.75, with softening."""
    rng = oracle.Rng(1)
    for s in (1, 2, 10, 100, 1000, 2000):
        m, x, y, z = rng.uniform_particles(s, 1.0, dtype)
        for max_leaf_n in (1, 2, 8, 16):
            for ncrit in (1, 16, 128, 256):
                ot = oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=max_leaf_n, ncrit=ncrit)
                st = state_from_oracle(ot)
                st.set_variant(variant)
                for theta, eps in ((0.001, 0.0), (0.75, 0.05)):
                    ref = ot.acc_pot(2, theta, eps=eps, nthreads=4)
                    got = st.acc_pot(2, mac_value_of(theta, "bh", dtype), eps2=float(dtype(eps) ** 2))
                    check(got, ref, 2, dtype, tol=TIGHT[dtype] * 5)


@pytest.mark.parametrize("variant", VARIANTS)
def test_big_groups_and_huge_leaves(variant):
    """Groups beyond 128/256/512 particles (ncrit up to 5000) and leaves far larger than an LDS tile
    (max_leaf_n 700; 1500 coincident particles in one deepest-level cell, eps > 0)."""
    rng = oracle.Rng(7)
    dtype = np.float32
    m, x, y, z = rng.uniform_particles(6000, 1.0, dtype)
    x[:1500], y[:1500], z[:1500] = 0.123, -0.2, 0.31  # one cell, 1500 particles
    for max_leaf_n, ncrit in ((16, 200), (16, 300), (16, 600), (700, 5000), (16, 5000)):
        ot = oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=max_leaf_n, ncrit=ncrit)
        st = state_from_oracle(ot)
        st.set_variant(variant)
        ref = ot.acc_pot(2, 0.6, eps=0.01, nthreads=8)
        got = st.acc_pot(2, mac_value_of(0.6, "bh", dtype), eps2=float(dtype(0.01) ** 2))
        check(got, ref, 2, dtype, tol=1e-4)


@pytest.mark.parametrize("dtype,ndim,mac", [(np.float64, 3, "bh"), (np.float32, 3, "bh_geom"), (np.float64, 2, "bh"),
                                            (np.float32, 2, "bh_geom")])
def test_chunked_big_groups_all_flavours(dtype, ndim, mac):
    """Critical nodes of more than 256 particles are cut into chunks of targets that share the node's MAC decisions
    (k_list<BIG>): every flavour against the oracle, Q = 0, 1, 2; the union of two sub-ranges equals the full-range
    result bit for bit; repeated calls give the same bits."""
    rng = oracle.Rng(21)
    n = 7000
    if ndim == 3:
        m, x, y, z = rng.uniform_particles(n, 1.0, dtype)
        x[:900], y[:900], z[:900] = 0.2, 0.1, -0.3  # 900 coincident particles: one deepest-level leaf
        ot = oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=300, ncrit=1300, mac=mac)
    else:
        m, x, y = rng.uniform_particles(n, 1.0, dtype, ndim=2)
        x[:900], y[:900] = 0.2, 0.1
        ot = oracle.Tree(x, y, None, m, box_size=1.0, max_leaf_n=300, ncrit=1300, mac=mac, ndim=2)
    st = state_from_oracle(ot)
    cr = st.crit_ranges()
    assert (cr[:, 1] - cr[:, 0]).max() > 600
    mv = mac_value_of(0.6, mac, dtype)
    cut = int(cr[len(cr) // 2, 0])
    for q in (0, 1, 2):
        ref = ot.acc_pot(q, 0.6, eps=0.01, nthreads=8)
        got = st.acc_pot(q, mv, eps2=float(dtype(0.01) ** 2))
        check(got, ref, q, dtype, tol=1e-4 if dtype == np.float32 else 1e-11, ndim=ndim)
        again = st.acc_pot(q, mv, eps2=float(dtype(0.01) ** 2))
        lo = st.acc_pot(q, mv, eps2=float(dtype(0.01) ** 2), p_begin=0, p_end=cut, offset_output=False)
        hi = st.acc_pot(q, mv, eps2=float(dtype(0.01) ** 2), p_begin=cut, p_end=n, offset_output=False)
        for a, b, l, h in zip(got, again, lo, hi):
            assert np.array_equal(a, b)
            assert np.array_equal(a, np.concatenate([l, h]))


@pytest.mark.parametrize("variant", VARIANTS)
def test_variants_agree_and_deterministic(variant):
    m, x, y, z = oracle.plummer(50000, np.float32)
    ot = oracle.Tree(x, y, z, m)
    st = state_from_oracle(ot)
    st.set_variant(variant)
    mv = mac_value_of(0.75, "bh", np.float32)
    a = st.acc_pot(2, mv)
    b = st.acc_pot(2, mv)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)  # bit-identical across calls (test/g_constant_acc.cpp:66-88)
    # Exact G scaling: G applied as the final multiply.
    c = st.acc_pot(2, mv, G=2.0)
    for u, v in zip(a, c):
        assert np.array_equal(u * np.float32(2), v)
    z0 = st.acc_pot(2, mv, G=0.0)
    for v in z0:
        assert np.all(v == 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_list_and_producer_consumer_kernels_give_the_same_bits(dtype):
    """The producer / consumer kernel reproduces the list kernel's tiles, lane mapping and summation order: identical
    results bit for bit, for every Q, with and without softening, on sub-ranges, for octrees and quadtrees -- which is
    what allows the automatic variant to pick a kernel per lane-mapping class and per call by launch size alone."""
    for n, ndim in ((40000, 3), (9000, 3), (20000, 2)):
        if ndim == 3:
            m, x, y, z = oracle.plummer(n, dtype)
            ot = oracle.Tree(x, y, z, m, mac="bh_geom" if n == 9000 else "bh")
        else:
            m, x, y = oracle.Rng(4).uniform_particles(n, 3.0, dtype, ndim=2)
            ot = oracle.Tree(x, y, None, m, ndim=2)
        st = state_from_oracle(ot)
        mv = mac_value_of(0.6, ot.mac, dtype)
        cr = st.crit_ranges()
        b, e = int(cr[len(cr) // 5, 0]), int(cr[4 * len(cr) // 5, 0])
        for q in (0, 1, 2):
            for eps2 in (0.0, 1e-5):
                res = {}
                for v in (2, 3, 0):
                    st.set_variant(v)
                    res[v] = (st.acc_pot(q, mv, eps2=eps2, G=1.25), st.acc_pot(q, mv, eps2=eps2, p_begin=b, p_end=e, offset_output=False))
                for v in (3, 0):
                    for full_a, full_b in zip(res[2][0], res[v][0]):
                        assert np.array_equal(full_a, full_b), (n, ndim, q, eps2, v)
                    for part_a, part_b in zip(res[2][1], res[v][1]):
                        assert np.array_equal(part_a, part_b), (n, ndim, q, eps2, v)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_accs_pots_equal_accs_and_pots(dtype, variant):
    """accs_u(), pots_u() and accs_pots_u() evaluate the same expressions (as the reference's batch_batch_3d_* do,
    tree.hpp:2008-2068): the accelerations of Q = 0 and the potentials of Q = 1 are those of Q = 2, bit for bit."""
    m, x, y, z = oracle.plummer(30000, dtype)
    st = state_from_oracle(oracle.Tree(x, y, z, m))
    st.set_variant(variant)
    mv = mac_value_of(0.7, "bh", dtype)
    both = st.acc_pot(2, mv, eps2=1e-5, G=0.75)
    accs = st.acc_pot(0, mv, eps2=1e-5, G=0.75)
    pots = st.acc_pot(1, mv, eps2=1e-5, G=0.75)
    for k in range(3):
        assert np.array_equal(accs[k], both[k])
    assert np.array_equal(pots[0], both[3])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_split_traversal_one_wave_per_node_or_per_part_same_bits(dtype, monkeypatch):
    """Variant 4 sums every target's contributions in an order that is a function of the node and the MAC value only (tiles
    of 128 sources, parts of four tiles added up in order): one wavefront per node, one wavefront per part (+ k_combine),
    and shards that run in the other form than the full range all give the same bits; octree and quadtree, Q = 0, 1, 2."""
    for n, ndim in ((40000, 3), (20000, 2)):
        if ndim == 3:
            m, x, y, z = oracle.plummer(n, dtype)
            ot = oracle.Tree(x, y, z, m)
        else:
            m, x, y = oracle.Rng(4).uniform_particles(n, 3.0, dtype, ndim=2)
            ot = oracle.Tree(x, y, None, m, ndim=2)
        st = state_from_oracle(ot)
        st.set_variant(4)
        mv = mac_value_of(0.6, ot.mac, dtype)
        cr = st.crit_ranges()
        b, e = int(cr[len(cr) // 5, 0]), int(cr[4 * len(cr) // 5, 0])
        for q in (0, 1, 2):
            monkeypatch.setenv("RK_SL_PARTS_BELOW", "0")
            whole = st.acc_pot(q, mv, eps2=1e-5, G=1.25)
            monkeypatch.setenv("RK_SL_PARTS_BELOW", "100000000")
            parts = st.acc_pot(q, mv, eps2=1e-5, G=1.25)
            monkeypatch.setenv("RK_SL_PARTS_BELOW", str(len(cr) // 2))  # the shard runs per part, the full range per node
            shard = st.acc_pot(q, mv, eps2=1e-5, G=1.25, p_begin=b, p_end=e, offset_output=False)
            for w, p, s in zip(whole, parts, shard):
                assert np.array_equal(w, p), (n, ndim, q)
                assert np.array_equal(w[b:e], s), (n, ndim, q)
