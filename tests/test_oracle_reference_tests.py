"""The reference's own known-answer tests for the acc/pot path, restated on the CPU oracle.

Each test cites the reference test it restates (paths relative to /root/reference/test). Inputs come
from the same generator (test_utils.hpp:41-59 restated in oracle.Rng) with the same seeds; bounds are
the reference's. The matrix of accuracy_*.cpp is thinned (same extremes) to keep the CPU suite short.
"""
import numpy as np
import pytest

import oracle

FP = [np.float32, np.float64]
MACS = ["bh", "bh_geom"]


def _tree(parts, dtype, **kw):
    m, x, y, z = parts
    return oracle.Tree(x, y, z, m, **kw)


@pytest.mark.parametrize("mac", MACS)
@pytest.mark.parametrize("dtype", FP)
@pytest.mark.parametrize("q", [0, 1, 2])
def test_accuracy_vs_exact(mac, dtype, q):
    """accuracy_acc.cpp:49-120, accuracy_pot.cpp, accuracy_acc_pot.cpp: theta=0.001, box=1, ordered and
    unordered; all finite; fp64 max relative error < 5e-10 (accs) / 1e-10 (pots)."""
    rng = oracle.Rng(1)
    theta, bsize = 0.001, 1.0
    tot = 0.0
    for s in (10, 100, 1000):
        parts = rng.uniform_particles(s, bsize, dtype)
        for max_leaf_n, ncrit in ((1, 1), (2, 16), (8, 128), (16, 256), (16, 1)):
            t = _tree(parts, dtype, box_size=bsize, max_leaf_n=max_leaf_n, ncrit=ncrit, mac=mac)
            for ordered in (True, False):
                res = t.acc_pot(q, theta, ordered=ordered)
                for r in res:
                    assert np.all(np.isfinite(r))
                idxs = range(s) if s <= 100 else range(0, s, 37)
                for i in idxs:
                    ex = t.exact(q, i, ordered=ordered)
                    got = np.array([r[i] for r in res], dtype=np.float64)
                    tot = max(tot, np.max(np.abs((ex - got) / ex)))
    if dtype == np.float64:
        assert tot < (5e-10 if q != 1 else 1e-10), tot


@pytest.mark.parametrize("mac", MACS)
@pytest.mark.parametrize("dtype", FP)
def test_g_constant(mac, dtype):
    """g_constant_acc.cpp:43-100 (and _pot, _acc_pot): theta=0.75, N=10000; G=0 gives exact zeros; G=2 and
    G=1/2 give bit-exact multiples of the G=1 result."""
    rng = oracle.Rng(0)
    parts = rng.uniform_particles(10000, 10.0, dtype)
    t = _tree(parts, dtype, box_size=10.0, mac=mac)
    for q in (0, 1, 2):
        base = t.acc_pot(q, 0.75, nthreads=8)
        for r in t.acc_pot(q, 0.75, G=0.0, nthreads=8):
            assert np.all(r == 0)
        for G in (2.0, 0.5):
            res = t.acc_pot(q, 0.75, G=G, nthreads=8)
            for r, b in zip(res, base):
                assert np.array_equal(r, b * dtype(G))
        again = t.acc_pot(q, 0.75, nthreads=3)
        for r, b in zip(again, base):
            assert np.array_equal(r, b)  # run-to-run determinism


@pytest.mark.parametrize("mac", MACS)
@pytest.mark.parametrize("dtype", FP)
def test_zero_masses(mac, dtype):
    """zero_masses.cpp:36-77: all masses zero, theta=0.75: every output finite and exactly zero."""
    rng = oracle.Rng(0)
    m, x, y, z = rng.uniform_particles(5000, 10.0, dtype)
    t = oracle.Tree(x, y, z, np.zeros_like(m), box_size=10.0, mac=mac)
    for q in (0, 1, 2):
        for ordered in (False, True):
            for r in t.acc_pot(q, 0.75, ordered=ordered, nthreads=8):
                assert np.all(np.isfinite(r)) and np.all(r == 0)


@pytest.mark.parametrize("mac", MACS)
@pytest.mark.parametrize("dtype", FP)
def test_softening(mac, dtype):
    """softening_acc.cpp:50-165 (and _pot, _acc_pot): eps in {0, 0.1, 100} against the softened direct sum,
    theta=0.001 (fp64 < 1e-10); coincident particles with eps > 0 stay finite (softening_acc.cpp:140-145)."""
    rng = oracle.Rng(1)
    s = 1000
    parts = rng.uniform_particles(s, 1.0, dtype)
    for eps in (0.0, 0.1, 100.0):
        t = _tree(parts, dtype, box_size=1.0, max_leaf_n=8, ncrit=16, mac=mac)
        for q in (0, 1, 2):
            res = t.acc_pot(q, 0.001, eps=eps, ordered=True)
            worst = 0.0
            for i in range(0, s, 29):
                ex = t.exact(q, i, eps=eps, ordered=True)
                got = np.array([r[i] for r in res], dtype=np.float64)
                worst = max(worst, np.max(np.abs((ex - got) / ex)))
            if dtype == np.float64:
                assert worst < 1e-10, worst
    m, x, y, z = (v.copy() for v in parts)
    x[:50], y[:50], z[:50] = x[50:100], y[50:100], z[50:100]  # coincident pairs
    t = oracle.Tree(x, y, z, m, box_size=1.0, mac=mac)
    for q in (0, 1, 2):
        for r in t.acc_pot(q, 0.75, eps=0.1):
            assert np.all(np.isfinite(r))


@pytest.mark.parametrize("dtype", FP)
def test_ordering(dtype):
    """ordering_acc.cpp:44-196: accs_o agrees with the direct sum in the ORIGINAL order, theta=0.01,
    N=10000, relative difference of |a| <= 2e-3 (fp32) / 2e-11 (fp64)."""
    rng = oracle.Rng(2)
    s = 10000
    m, x, y, z = rng.uniform_particles(s, 1.0, dtype)
    t = oracle.Tree(x, y, z, m, box_size=1.0)
    res = t.accs_o(0.01, nthreads=8)
    tol = 2e-3 if dtype == np.float32 else 2e-11
    cp = t.codes_perms()
    xs, ys, zs, ms = t.parts_u()
    for i in range(0, s, 997):
        # The particle at original index i sits at Morton index inv_perm[i].
        j = int(cp["inv_perm"][i])
        assert xs[j] == x[i] and ys[j] == y[i] and zs[j] == z[i] and ms[j] == m[i]
        ex = t.exact(0, i, ordered=True).astype(np.float64)
        got = np.array([r[i] for r in res], dtype=np.float64)
        ne, ng = np.linalg.norm(ex), np.linalg.norm(got)
        assert abs(ne - ng) / ne <= tol


def test_error_behaviour():
    """Domain errors of acc_pot_dispatch (tree.hpp:3299-3319) and constructor errors (tree.hpp:1350-1362)."""
    rng = oracle.Rng(0)
    m, x, y, z = rng.uniform_particles(100, 1.0, np.float64)
    t = oracle.Tree(x, y, z, m, box_size=1.0)
    with pytest.raises(ArithmeticError, match="MAC value must be finite and positive"):
        t.accs_u(0.0)
    with pytest.raises(ArithmeticError, match="softening length must be finite and non-negative"):
        t.accs_u(0.5, eps=-1.0)
    with pytest.raises(ArithmeticError, match="gravitational constant G must be finite"):
        t.accs_u(0.5, G=float("inf"))
    with pytest.raises(ValueError, match="maximum number of particles per leaf must be nonzero"):
        oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=0)
    with pytest.raises(ValueError, match="outside the allowed bounds"):
        oracle.Tree(x, y, z, m, box_size=0.5)
