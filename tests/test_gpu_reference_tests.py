"""The reference's acc/pot tests restated on the PRODUCT (rakau_amd.Octree -> C ABI -> HIP kernels).
Each test cites the reference test it mirrors (paths relative to /root/reference/test)."""
import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import rel_err_vec, rel_err

pytestmark = pytest.mark.gpu
FP = [np.float32, np.float64]
MACS = ["bh", "bh_geom"]


@pytest.mark.parametrize("mac", MACS)
@pytest.mark.parametrize("dtype", FP)
def test_accuracy_vs_exact(mac, dtype):
    """accuracy_acc.cpp:49-120 / accuracy_pot.cpp / accuracy_acc_pot.cpp: theta = 0.001, unit box, ordered
    and unordered, all finite, fp64 per-component max relative error < 5e-10 (accs), 1e-10 (pots)."""
    rng = oracle.Rng(1)
    worst_acc = worst_pot = 0.0
    for s in (10, 100, 1000, 2000):
        m, x, y, z = rng.uniform_particles(s, 1.0, dtype)
        for max_leaf_n, ncrit in ((1, 1), (2, 16), (8, 128), (16, 256), (16, 1)):
            t = rakau_amd.Octree(x, y, z, m, box_size=1.0, max_leaf_n=max_leaf_n, ncrit=ncrit, mac=mac)
            for ordered in (True, False):
                res = t.accs_pots_o(0.001) if ordered else t.accs_pots_u(0.001)
                for r in res:
                    assert np.all(np.isfinite(r))
                for i in (range(s) if s <= 100 else range(0, s, 41)):
                    ex = (t.exact_acc_pot_o(i) if ordered else t.exact_acc_pot_u(i)).astype(np.float64)
                    got = np.array([r[i] for r in res], dtype=np.float64)
                    e = np.abs((ex - got) / ex)
                    worst_acc, worst_pot = max(worst_acc, e[:3].max()), max(worst_pot, e[3])
    if dtype == np.float64:
        assert worst_acc < 5e-10 and worst_pot < 1e-10, (worst_acc, worst_pot)
    else:
        assert worst_acc < 5e-2 and worst_pot < 1e-4, (worst_acc, worst_pot)  # the reference sets no fp32 bound


@pytest.mark.parametrize("mac", MACS)
@pytest.mark.parametrize("dtype", FP)
def test_g_constant(mac, dtype):
    """g_constant_acc.cpp:43-100: theta = 0.75, N = 10000; G = 0 -> exact zeros; G = 2 and 1/2 -> bit-exact
    multiples of the G = 1 result (also for pots and accs+pots)."""
    rng = oracle.Rng(0)
    m, x, y, z = rng.uniform_particles(10000, 10.0, dtype)
    t = rakau_amd.Octree(x, y, z, m, box_size=10.0, mac=mac)
    for fn in ("accs_u", "accs_pots_u", "accs_o"):
        base = getattr(t, fn)(0.75)
        for r in getattr(t, fn)(0.75, G=0.0):
            assert np.all(r == 0)
        for G in (2.0, 0.5):
            for r, b in zip(getattr(t, fn)(0.75, G=G), base):
                assert np.array_equal(r, b * dtype(G))
    p = t.pots_u(0.75)
    assert np.array_equal(t.pots_u(0.75, G=2.0), p * dtype(2))


@pytest.mark.parametrize("mac", MACS)
@pytest.mark.parametrize("dtype", FP)
def test_zero_masses(mac, dtype):
    """zero_masses.cpp:36-77: all six entry points return finite exact zeros."""
    rng = oracle.Rng(0)
    m, x, y, z = rng.uniform_particles(5000, 10.0, dtype)
    t = rakau_amd.Octree(x, y, z, np.zeros_like(m), box_size=10.0, mac=mac)
    outs = t.accs_u(0.75) + t.accs_o(0.75) + [t.pots_u(0.75), t.pots_o(0.75)] + t.accs_pots_u(0.75) + t.accs_pots_o(0.75)
    for r in outs:
        assert np.all(np.isfinite(r)) and np.all(r == 0)


@pytest.mark.parametrize("dtype", FP)
def test_softening(dtype):
    """softening_acc.cpp:50-165: eps in {0, 0.1, 100} vs the softened direct sum at theta = 0.001 (fp64 < 1e-10);
    coincident particles with eps > 0 stay finite (softening_acc.cpp:140-145)."""
    rng = oracle.Rng(1)
    s = 1000
    m, x, y, z = rng.uniform_particles(s, 1.0, dtype)
    for eps in (0.0, 0.1, 100.0):
        t = rakau_amd.Octree(x, y, z, m, box_size=1.0, max_leaf_n=8, ncrit=16)
        res = t.accs_pots_o(0.001, eps=eps)
        worst = 0.0
        for i in range(0, s, 29):
            ex = t.exact_acc_pot_o(i, eps=eps).astype(np.float64)
            got = np.array([r[i] for r in res], dtype=np.float64)
            worst = max(worst, np.max(np.abs((ex - got) / ex)))
        assert worst < (1e-10 if dtype == np.float64 else 5e-3), worst
    x2, y2, z2 = x.copy(), y.copy(), z.copy()
    x2[:50], y2[:50], z2[:50] = x[50:100], y[50:100], z[50:100]
    t = rakau_amd.Octree(x2, y2, z2, m, box_size=1.0)
    for r in t.accs_pots_u(0.75, eps=0.1):
        assert np.all(np.isfinite(r))


@pytest.mark.parametrize("dtype", FP)
def test_ordering(dtype):
    """ordering_acc.cpp:44-196: accs_o in the ORIGINAL order agrees with the direct sum, theta = 0.01, N = 10000,
    |a| within 2e-3 (fp32) / 2e-11 (fp64); still true after update_particles_u (a rotation)."""
    rng = oracle.Rng(2)
    s = 10000
    m, x, y, z = rng.uniform_particles(s, 1.0, dtype)
    t = rakau_amd.Octree(x, y, z, m, box_size=4.0)
    tol = 2e-3 if dtype == np.float32 else 2e-11

    def check():
        res = t.accs_o(0.01)
        for i in range(0, s, 997):
            ex = t.exact_acc_o(i).astype(np.float64)
            got = np.array([r[i] for r in res], dtype=np.float64)
            assert abs(np.linalg.norm(ex) - np.linalg.norm(got)) / np.linalg.norm(ex) <= tol

    check()
    c, sn = dtype(np.cos(0.7)), dtype(np.sin(0.7))

    def rot(a):
        ax, ay = a[0].copy(), a[1].copy()
        a[0][:], a[1][:] = c * ax - sn * ay, sn * ax + c * ay

    t.update_particles_u(rot)
    check()


def test_split_and_errors():
    """Error behaviour of the dispatch (tree.hpp:2857-2868, 3136-3141, 3299-3319) and split semantics."""
    rng = oracle.Rng(3)
    m, x, y, z = rng.uniform_particles(5000, 1.0, np.float32)
    t = rakau_amd.Octree(x, y, z, m, box_size=1.0)
    base = t.accs_u(0.75)
    cpu = t.cpu_acc_pot_u(0, 0.75)
    # split = {cpu, dev0} (tree.hpp:3047-3113): the CPU engine computes the critical nodes below the cut while the GPU
    # computes the rest. Every particle's result is the one its engine gives alone, bit for bit (critical nodes are
    # independent); the two engines agree with each other to rounding (same interaction lists).
    assert max(float(np.abs(c - b).max() / np.abs(b).max()) for c, b in zip(cpu, base)) < 1e-5
    for r, b in zip(t.accs_u(0.75, split=(0.0, 1.0)), base):
        assert np.array_equal(r, b)
    for sp in ((1.0, 0.0), (0.3,), (1.0, 1e-9)):  # no device share, CPU only, device share below rk_min_size()
        for r, c in zip(t.accs_u(0.75, split=sp), cpu):
            assert np.array_equal(r, c)
    crit = t.crit_nodes()
    for sp in ((0.5, 0.5), (0.2, 0.8), (3.0, 1.0)):
        want = int(sp[0] / (sp[0] + sp[1]) * t.nparts)
        i = int(np.searchsorted(crit[:, 1].astype(np.int64), want, side="left"))
        cut = int(crit[i, 1]) if i < len(crit) else t.nparts  # snapped to a critical node (tree.hpp:3053-3063)
        assert 0 < cut < t.nparts
        for r, c, b in zip(t.accs_pots_u(0.75, split=sp, eps=1e-3, G=2.0), t.cpu_acc_pot_u(2, 0.75, eps=1e-3, G=2.0),
                           t.accs_pots_u(0.75, eps=1e-3, G=2.0)):
            assert np.array_equal(r[:cut], c[:cut]) and np.array_equal(r[cut:], b[cut:])
    # _o outputs of a split call: the same values scattered through perm.
    perm = t.perm().astype(np.int64)
    for u, o in zip(t.accs_u(0.75, split=(0.5, 0.5)), t.accs_o(0.75, split=(0.5, 0.5))):
        assert np.array_equal(o[perm], u)
    with pytest.raises(ValueError, match="cannot contain non-finite"):
        t.accs_u(0.75, split=(float("nan"), 1.0))
    with pytest.raises(ValueError, match="only non-negative"):
        t.accs_u(0.75, split=(-1.0, 1.0))
    with pytest.raises(ValueError, match="cannot all be zero"):
        t.accs_u(0.75, split=(0.0, 0.0))
    with pytest.raises(ValueError, match="accelerators, but only"):
        t.accs_u(0.75, split=(1.0,) * 40)
    with pytest.raises(ArithmeticError, match="MAC value must be finite and positive"):
        t.accs_u(-1.0)
    with pytest.raises(ArithmeticError, match="softening length must be finite and non-negative"):
        t.accs_u(0.75, eps=-0.1)
    with pytest.raises(ArithmeticError, match="gravitational constant G must be finite"):
        t.accs_u(0.75, G=float("nan"))


def test_state_range_and_replication():
    """rk_acc_pot on sub-ranges (critical-node aligned), compact vs offset outputs, misaligned ranges rejected,
    and rk_state_export/import (the replication used for multi-GPU runs) on the same device."""
    m, x, y, z = oracle.plummer(30000, np.float32)
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    full = st.acc_pot(0, mv)
    cr = st.crit_ranges()
    mid = int(cr[len(cr) // 2, 0])
    lo = st.acc_pot(0, mv, p_begin=0, p_end=mid)
    hi = st.acc_pot(0, mv, p_begin=mid, p_end=None, offset_output=False)
    for k in range(3):
        assert np.array_equal(lo[k][:mid], full[k][:mid]) and np.all(lo[k][mid:] == 0)
        assert np.array_equal(hi[k], full[k][mid:])
    with pytest.raises(ValueError, match="critical node boundaries"):
        big = int(np.argmax(cr[:, 1] - cr[:, 0] > 1))  # a group with at least two particles
        st.acc_pot(0, mv, p_begin=int(cr[big, 0]) + 1)
    ptrs, nbytes, meta = st.export()
    clone = rakau_amd.State.from_buffers(0, ptrs, nbytes, meta)
    for a, b in zip(clone.acc_pot(0, mv), full):
        assert np.array_equal(a, b)
    c = st.count_interactions(mv)
    _, stats = oracle.Tree(x, y, z, m).acc_pot(0, 0.75, nthreads=8, want_stats=True)
    assert (c["mac"], c["com"], c["pp"], c["self"]) == (stats["w_visits"], stats["w_com"], stats["w_pp"], stats["w_self"])


@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_median_error(mac, dtype):
    """median_error_acc.cpp:44-80: 5000 uniform particles in a box of 100, theta in {0.2, 0.4, 0.6, 0.8}; the reference
    only prints the median of |a_tree - a_exact| / |a_exact|. Here: the medians grow with theta, stay in the Barnes-Hut
    range, and equal the CPU oracle's medians for the same tree to 1 % (same interaction lists)."""
    rng = oracle.Rng(1)
    n = 5000
    m, x, y, z = rng.uniform_particles(n, 100.0, dtype)
    t = rakau_amd.Octree(x, y, z, m, mac=mac)
    ot = oracle.Tree(x, y, z, m, mac=mac)
    ex = np.stack([t.exact_acc_u(i).astype(np.float64) for i in range(0, n, 5)])
    meds = []
    for theta in (0.2, 0.4, 0.6, 0.8):
        acc = np.stack(t.accs_u(theta, split=[0.5, 0.5]), axis=1).astype(np.float64)[::5]
        ref = np.stack(ot.acc_pot(0, theta, nthreads=4), axis=1).astype(np.float64)[::5]
        err = np.linalg.norm(acc - ex, axis=1) / np.linalg.norm(ex, axis=1)
        err_o = np.linalg.norm(ref - ex, axis=1) / np.linalg.norm(ex, axis=1)
        meds.append(float(np.median(err)))
        assert abs(np.median(err) - np.median(err_o)) <= 0.01 * np.median(err_o) + 1e-7
    print("\nmedian errors (theta 0.2, 0.4, 0.6, 0.8), mac=%s, %s: %s" % (mac, np.dtype(dtype).name, ["%.2e" % v for v in meds]))
    assert all(a < b for a, b in zip(meds, meds[1:])) and meds[0] < 2e-3 and meds[-1] < 3e-2
