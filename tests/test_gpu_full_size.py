"""BASELINE.json full-size configurations through the product: size-independent properties (finite, deterministic,
exact G scaling, shard union == full result bit for bit, direct sums) plus the oracle on bounded samples of critical
nodes taken from the start, the quartiles and the end of the Morton order and around the innermost and outermost
particle (core and halo of the Plummer sphere).

  config 2: 4M fp32 theta=0.75 accs_u                     -> test_4m_accs_u_headline (Q = 0, default variant, seven windows),
                                                             test_4m_properties (Q = 2, variants 0 and 1)
  config 3: 4M fp32 theta=0.75 accs_pots_u with softening -> test_4m_accs_pots_softened
  config 4: 16M fp64 theta=0.5 accs_u                     -> test_16m_fp64_theta05
  config 5: 64M fp32 theta=0.75, whole and as 8 Morton shards run back to back -> test_64m_sharded
"""
import gc

import numpy as np
import pytest

import oracle
import rakau_amd
from bench import plummer_numpy, shard_cuts, usable_cpus

pytestmark = pytest.mark.gpu


def sample_windows(ot, width):
    """Contiguous windows [c_begin, c_end) of critical nodes spread over the Morton order, plus the windows around
    the critical nodes that hold the particle nearest to and farthest from the centre of the sphere."""
    crit = ot.crit_nodes()
    ng = len(crit)
    px, py, pz, _ = ot.parts_u()
    r2 = px.astype(np.float64) ** 2 + py.astype(np.float64) ** 2 + pz.astype(np.float64) ** 2
    begins = crit[:, 1].astype(np.int64)
    centres = [0, ng // 4, ng // 2, 3 * ng // 4, ng - 1,
               int(np.searchsorted(begins, int(np.argmin(r2)), side="right")) - 1,
               int(np.searchsorted(begins, int(np.argmax(r2)), side="right")) - 1]
    del px, py, pz, r2
    wins = []
    for c in centres:
        a = max(0, min(c - width // 2, ng - width))
        wins.append((a, min(ng, a + width)))
    return crit, sorted(set(wins))


def check_full_size(n, dtype, theta, q, eps, n_shards, width, tol_max, tol_med, n_direct):
    """One BASELINE configuration through the product's builder and the C ABI, checked as the module docstring says.
    Tolerances are on |a - a_ref| / |a_ref| per particle (vector norm, SURVEY 8(d) Gate A) and on the relative
    potential error; the reference's own bounds are 2e-3 (fp32) and 2e-11 (fp64), test/ordering_acc.cpp:94-96."""
    m, x, y, z = plummer_numpy(n, dtype)
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(theta, "bh", dtype)
    f = np.dtype(dtype).type
    eps2 = float(f(eps) ** 2)
    nacc = 3 if q != 1 else 0
    full = st.acc_pot(q, mv, eps2=eps2)
    for r in full:
        assert np.all(np.isfinite(r))
    # Determinism and exact G scaling at full size (test/g_constant_acc.cpp:66-88 of the reference).
    again = st.acc_pot(q, mv, eps2=eps2, G=2.0)
    for u, v in zip(full, again):
        assert np.array_equal(u * f(2), v)
    del again
    # Morton shards cut at critical-node boundaries into equal work reproduce the full result bit for bit.
    cuts = shard_cuts(st.crit_ranges(), n, n_shards, st.group_work(mv))
    assert cuts[0] == 0 and cuts[-1] == n and all(b > a for a, b in zip(cuts, cuts[1:]))
    for r in range(n_shards):
        part = st.acc_pot(q, mv, eps2=eps2, p_begin=cuts[r], p_end=cuts[r + 1], offset_output=False)
        for u, v in zip(part, full):
            assert np.array_equal(u, v[cuts[r]:cuts[r + 1]]), "shard %d differs from the full result" % r
        del part
    # Direct sums on a few particles (independent of the oracle): BH truncation at this theta stays below 5e-2
    # (the reference's CPU engine shows ~2e-2 worst case at theta = 0.75).
    xs, ys, zs, ms = t.p_its_u()
    idx = np.random.default_rng(1).choice(n, n_direct, replace=False)
    for i in idx:
        acc = np.zeros(3)
        pot = 0.0
        for b in range(0, n, 1 << 22):
            e = min(n, b + (1 << 22))
            d = np.stack([xs[b:e], ys[b:e], zs[b:e]], 1).astype(np.float64) - [xs[i], ys[i], zs[i]]
            r2 = (d * d).sum(1) + eps2
            if b <= i < e:
                r2[i - b] = np.inf
            w = ms[b:e] / r2 ** 1.5
            acc += (w[:, None] * d).sum(0)
            pot -= float(ms[i]) * (ms[b:e] / np.sqrt(r2)).sum()
        if nacc:
            got = np.array([full[k][i] for k in range(3)], dtype=np.float64)
            assert np.linalg.norm(got - acc) / np.linalg.norm(acc) < 5e-2
        if q != 0:
            assert abs(float(full[-1][i]) - pot) / abs(pot) < 5e-2
    del xs, ys, zs, ms
    # The oracle (same tree parameters; its tree equals the product's node for node) on the sampled windows.
    ot = oracle.Tree(x, y, z, m)
    assert (ot.n_nodes, ot.n_crit) == (t.n_nodes, t.n_crit)
    crit, wins = sample_windows(ot, width)
    assert np.array_equal(crit[:, 1:].astype(np.int64), st.crit_ranges())
    thr = usable_cpus()
    worst = worst_pot = 0.0
    for a, b in wins:
        lo, hi = int(crit[a, 1]), int(crit[b - 1, 2])
        ref = ot.acc_pot(q, theta, eps=eps, nthreads=thr, c_begin=a, c_end=b)
        if nacc:
            g = np.stack([v[lo:hi] for v in full[:3]], 1).astype(np.float64)
            r = np.stack([v[lo:hi] for v in ref[:3]], 1).astype(np.float64)
            e = np.linalg.norm(g - r, axis=1) / np.linalg.norm(r, axis=1)
            assert e.max() < tol_max and np.median(e) < tol_med, (a, b, e.max(), np.median(e))
            worst = max(worst, float(e.max()))
        if q != 0:
            pe = np.abs(full[-1][lo:hi].astype(np.float64) - ref[-1][lo:hi]) / np.abs(ref[-1][lo:hi])
            assert pe.max() < tol_max, (a, b, pe.max())
            worst_pot = max(worst_pot, float(pe.max()))
        del ref
    print("full size n=%d %s theta=%g q=%d: %d windows of %d critical nodes, worst rel err acc %.3g pot %.3g"
          % (n, np.dtype(dtype).name, theta, q, len(wins), width, worst, worst_pot))
    del ot, st, t, full
    gc.collect()


def test_4m_accs_u_headline():
    """BASELINE config 2, the configuration the metric is quoted on: accs_u() (Q = 0), unsoftened, default kernel variant,
    compared with the oracle on the same seven windows as configs 3-5 (start, quartiles, end, core, halo)."""
    check_full_size(4_000_000, "float32", 0.75, 0, 0.0, n_shards=8, width=600, tol_max=2e-4, tol_med=3e-6,
                    n_direct=20)


def test_4m_accs_pots_softened():
    """BASELINE config 3. The reference's only in-repo softening choice for this workload:
    eps = 0.45 * N^-0.73 (benchmark/benchmark_leapfrog.cpp:218-223)."""
    n = 4_000_000
    check_full_size(n, "float32", 0.75, 2, 0.45 * n ** -0.73, n_shards=8, width=600, tol_max=2e-4, tol_med=3e-6,
                    n_direct=20)


def test_16m_fp64_theta05():
    """BASELINE config 4: tighter MAC, deeper stacks, the fp64 kernels (96-byte node records, 4 waves per SIMD)."""
    check_full_size(16_000_000, "float64", 0.5, 0, 0.0, n_shards=8, width=300, tol_max=1e-11, tol_med=1e-14,
                    n_direct=6)


def test_64m_sharded():
    """BASELINE config 5 on one GPU: the whole 64M problem, then as 8 Morton shards run back to back (what the 8
    GPUs of the sharded run compute, one shard each, on the replicated tree); the union is bit-identical."""
    check_full_size(64_000_000, "float32", 0.75, 0, 0.0, n_shards=8, width=600, tol_max=2e-4, tol_med=1e-5,
                    n_direct=3)


def test_4m_properties():
    n = 4_000_000
    m, x, y, z = plummer_numpy(n, "float32")
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    a = st.acc_pot(2, mv)
    for r in a:
        assert np.all(np.isfinite(r))
    # Determinism and exact G scaling at full size.
    b = st.acc_pot(2, mv, G=2.0)
    for u, v in zip(a, b):
        assert np.array_equal(u * np.float32(2), v)
    # Shards of a 4-way Morton split reproduce the full result bit for bit (no cross-shard reduction).
    cuts = shard_cuts(st.crit_ranges(), n, 4)
    for r in range(4):
        part = st.acc_pot(2, mv, p_begin=cuts[r], p_end=cuts[r + 1], offset_output=False)
        for u, v in zip(part, a):
            assert np.array_equal(u, v[cuts[r]:cuts[r + 1]])
    # Both kernel variants agree to rounding (same interaction lists, different summation order).
    st.set_variant(1)
    c = st.acc_pot(0, mv)
    st.set_variant(0)
    g = np.stack(a[:3], 1).astype(np.float64)
    h = np.stack(c, 1).astype(np.float64)
    err = np.linalg.norm(g - h, axis=1) / np.linalg.norm(h, axis=1)
    # Rounding-level differences grow with cancellation in |a|; the reference's own fp32 bound is 2e-3
    # (test/ordering_acc.cpp:96). Observed here: ~3e-5 worst case over 4M particles.
    assert err.max() < 2e-4 and np.median(err) < 1e-6, (err.max(), np.median(err))
    # Direct-sum check of the potential energy identity on a sample: pot_i == -sum_j m_i m_j / r_ij within the
    # BH truncation error at theta = 0.75 (the reference's CPU engine shows ~2e-2 worst case).
    xs, ys, zs, ms = t.p_its_u()
    idx = np.random.default_rng(1).choice(n, 40, replace=False)
    P = np.stack([xs, ys, zs], 1).astype(np.float64)
    for i in idx:
        d = P - P[i]
        r2 = (d * d).sum(1)
        r2[i] = np.inf
        ex_acc = (ms[:, None] * d / r2[:, None] ** 1.5).sum(0)
        got = g[i]
        assert np.linalg.norm(got - ex_acc) / np.linalg.norm(ex_acc) < 5e-2
    # Oracle on the first 3000 critical nodes (same tree parameters): tight agreement.
    ot = oracle.Tree(x, y, z, m)
    crit = ot.crit_nodes()
    upto = int(crit[2999, 2])
    ref = ot.acc_pot(2, 0.75, nthreads=16, c_begin=0, c_end=3000)
    gr = np.stack([v[:upto] for v in ref[:3]], 1).astype(np.float64)
    e = np.linalg.norm(g[:upto] - gr, axis=1) / np.linalg.norm(gr, axis=1)
    assert e.max() < 2e-4 and np.median(e) < 1e-6, (e.max(), np.median(e))
    assert np.max(np.abs(a[3][:upto].astype(np.float64) - ref[3][:upto]) / np.abs(ref[3][:upto])) < 2e-5
