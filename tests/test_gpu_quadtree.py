"""2-dimensional (quadtree) variant behind the same boundary (the NDim = 2 instantiations
of the reference's accelerator seam, src/rakau_rocm.cpp:333-345): traversal parity against the 2-D oracle on identical
trees, the device builder against the oracle's tree, and the reference's accuracy properties on the product."""
import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import rel_err, rel_err_vec, state_from_oracle

pytestmark = pytest.mark.gpu


# fp32: a uniform sheet makes the net force a small difference of large sums, so rounding shows up at 1e-4 relative on a
# few particles (median 1e-7); still 20x inside the reference's own fp32 bound of 2e-3 (ordering_acc.cpp:93-97).
@pytest.mark.parametrize("dtype,tol,med", [(np.float32, 5e-4, 1e-6), (np.float64, 1e-12, 1e-14)])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
def test_traversal_parity_on_oracle_trees(dtype, tol, med, mac):
    rng = oracle.Rng(7)
    for s, max_leaf_n, ncrit, theta in ((3000, 16, 128, 0.75), (3000, 2, 16, 0.4), (500, 8, 256, 0.75), (17, 1, 1, 0.75),
                                        (6000, 16, 1000, 0.6)):
        m, x, y = rng.uniform_particles(s, 10.0, dtype, ndim=2)
        ot = oracle.Tree(x, y, None, m, box_size=10.0, max_leaf_n=max_leaf_n, ncrit=ncrit, mac=mac, ndim=2)
        st = state_from_oracle(ot)
        assert st.ndim == 2 and (st.nparts, st.tree_size, st.n_crit) == (ot.nparts, ot.n_nodes, ot.n_crit)
        mv = rakau_amd.mac_value_of(theta, mac, dtype)
        # The census (which nodes are accepted / opened for which group) is identical.
        _, stats = ot.acc_pot(0, theta, want_stats=True, nthreads=4)
        cen = st.count_interactions(mv)
        assert (cen["mac"], cen["com"], cen["pp"], cen["self"]) == (stats["w_visits"], stats["w_com"], stats["w_pp"],
                                                                    stats["w_self"])
        for variant in (0, 1):
            st.set_variant(variant)
            for q in (0, 1, 2):
                got = st.acc_pot(q, mv, G=1.5, eps2=1e-4)
                ref = ot.acc_pot(q, theta, G=1.5, eps=1e-2, nthreads=4)
                assert len(got) == len(ref) == rakau_amd.nres(q, 2)
                if q != 1:
                    e = rel_err_vec(got, ref, ndim=2)
                    assert e.max() <= tol and np.median(e) <= med
                if q != 0:
                    assert rel_err(got[-1], ref[-1]).max() <= tol


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
def test_device_built_quadtree(dtype, mac):
    rng = oracle.Rng(8)
    eps = np.finfo(dtype).eps
    for s in (1, 5, 300, 20000):
        m, x, y = rng.uniform_particles(s, 1.0, dtype, ndim=2)
        for max_leaf_n, ncrit, box in ((16, 128, None), (1, 1, 1.0), (8, 64, 3.0), (100, 300, None)):
            ot = oracle.Tree(x, y, None, m, box_size=box or 0.0, max_leaf_n=max_leaf_n, ncrit=ncrit, mac=mac, ndim=2)
            st = rakau_amd.State.build(x, y, None, m, box_size=box, max_leaf_n=max_leaf_n, ncrit=ncrit, mac=mac)
            assert st.ndim == 2 and st.tree_info()["box_size"] == ot.box_size
            assert (st.nparts, st.tree_size, st.n_crit) == (ot.nparts, ot.n_nodes, ot.n_crit)
            cp = ot.codes_perms()
            assert np.array_equal(st.download("codes"), cp["codes"]) and np.array_equal(st.download("perm"), cp["perm"])
            for k, ref in zip("xym", ot.parts_u()):
                assert np.array_equal(st.download(k), ref)
            with pytest.raises(ValueError, match="no z coordinates"):
                st.download("z")
            assert np.array_equal(st.download("crit"), ot.crit_nodes())
            dn, on = st.download("nodes"), ot.nodes()
            for k in ("begin", "end", "n_children", "code", "level"):
                assert np.array_equal(dn[k], on[k]), k
            assert np.array_equal(dn["dim2" if mac == "bh" else "dim"], on["dims"][:, 0])
            mo = on["props"][:, 2].astype(np.float64)
            assert np.max(np.abs(dn["props"][:, 2].astype(np.float64) - mo) / np.maximum(mo, 1e-300)) < 64 * eps
            assert np.abs(dn["props"][:, :2].astype(np.float64) - on["props"][:, :2]).max() < 256 * eps * ot.box_size
            if s >= 300:
                mv = rakau_amd.mac_value_of(0.75, mac, dtype)
                got, ref = st.acc_pot(2, mv), ot.acc_pot(2, 0.75, nthreads=4)
                tol = 2e-3 if dtype == np.float32 else 2e-11
                assert rel_err_vec(got, ref, ndim=2).max() <= tol and rel_err(got[2], ref[2]).max() <= tol


def test_deep_quadtree_beyond_21_levels():
    """Quadtrees go 31 levels deep (octrees 21): a tight clump at 1e-8 of the box needs levels 22-27."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.5, 0.5, 2000), 0.25 + rng.uniform(0, 1e-8, 600)])
    y = np.concatenate([rng.uniform(-0.5, 0.5, 2000), -0.125 + rng.uniform(0, 1e-8, 600)])
    m = rng.uniform(0.5, 1.5, 2600)
    ot = oracle.Tree(x, y, None, m, box_size=1.0, ndim=2)
    assert ot.nodes()["level"].max() > 21
    st = rakau_amd.State.build(x, y, None, m, box_size=1.0)
    assert np.array_equal(st.download("nodes")["code"], ot.nodes()["code"])
    assert np.array_equal(st.download("crit"), ot.crit_nodes())
    mv = rakau_amd.mac_value_of(0.5, "bh", np.float64)
    # The clump's particles sit 1e-8 apart: their mutual forces (~1e16) cancel to ~1e4, so rounding is amplified by
    # ~1e8 relative to the 3-D parity tests. Same tree (host-created state): summation order only.
    ref = ot.acc_pot(0, 0.5, nthreads=4)
    hs = state_from_oracle(ot)
    e = rel_err_vec(hs.acc_pot(0, mv), ref, ndim=2)
    assert e.max() < 1e-6 and np.median(e) < 1e-14
    e = rel_err_vec(st.acc_pot(0, mv), ref, ndim=2)
    assert e.max() < 1e-6 and np.median(e) < 1e-14


def test_device_outputs_ordered_and_rebuild_2d():
    import torch
    dev = torch.device("cuda", 0)
    rng = oracle.Rng(9)
    m, x, y = rng.uniform_particles(30000, 2.0, np.float32, ndim=2)
    ts = [torch.as_tensor(v).to(dev) for v in (x, y, m)]
    torch.cuda.synchronize()
    st = rakau_amd.State.build_device([t.data_ptr() for t in ts], 30000, np.float32)
    assert st.ndim == 2
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    ref = st.acc_pot(2, mv)
    perm = st.download("perm").astype(np.int64)
    outs = [torch.zeros(30000, dtype=torch.float32, device=dev) for _ in range(3)]
    st.acc_pot_device(2, mv, [o.data_ptr() for o in outs], ordered=True)
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        assert np.array_equal(o.cpu().numpy()[perm], r)
    ts[0].mul_(0.5)
    torch.cuda.synchronize()
    st.rebuild_device([t.data_ptr() for t in ts])
    fresh = rakau_amd.State.build(x * np.float32(0.5), y, None, m)
    assert st.ndim == 2 and np.array_equal(st.download("codes"), fresh.download("codes"))
    for a, b in zip(st.acc_pot(0, mv), fresh.acc_pot(0, mv)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("builder", ["host", "device"])
def test_quadtree_front_door(dtype, builder):
    """rakau::quadtree's acc/pot surface through the C++ header (Quadtree): ordering_acc.cpp / accuracy_acc.cpp style
    checks against the direct sum, _u vs _o, split invariance, G scaling, update_particles."""
    rng = oracle.Rng(21)
    s = 8000
    m, x, y = rng.uniform_particles(s, 1.0, dtype, ndim=2)
    t = rakau_amd.Quadtree(x, y, m, box_size=4.0, builder=builder)
    tol = 2e-3 if dtype == np.float32 else 2e-11
    au, ao = t.accs_u(0.01), t.accs_o(0.01)
    assert len(au) == 2 and len(t.accs_pots_o(0.75)) == 3
    perm = t.perm().astype(np.int64)
    for u, o in zip(au, ao):
        assert np.array_equal(o[perm], u)
    for i in range(0, s, 499):
        ex = t.exact_acc_o(i).astype(np.float64)
        got = np.array([a[i] for a in ao], dtype=np.float64)
        assert abs(np.linalg.norm(ex) - np.linalg.norm(got)) / np.linalg.norm(ex) <= tol
        assert abs(t.exact_pot_o(i, G=2.0, eps=0.1) - t.pots_o(0.01, G=2.0, eps=0.1)[i]) <= tol * abs(t.exact_pot_o(i, G=2.0, eps=0.1))
    a1, a2, a3 = t.accs_u(0.75), t.accs_u(0.75, G=2.0), t.accs_u(0.75, split=[0.3, 0.7])
    for u, v, w in zip(a1, a2, a3):
        # split = {cpu, dev0}: the first 30 % (snapped to a critical node) come from the CPU engine (rounding-level
        # agreement), the rest from the GPU (identical).
        assert np.array_equal(2 * u, v) and np.array_equal(u[s // 2:], w[s // 2:])
        assert np.abs(u - w).max() <= (1e-5 if dtype == np.float32 else 1e-13) * np.abs(u).max()
    ot = oracle.Tree(x, y, None, m, box_size=4.0, ndim=2)
    e = rel_err_vec(a1, ot.acc_pot(0, 0.75, nthreads=4), ndim=2)
    assert np.median(e) < (1e-6 if dtype == np.float32 else 1e-14) and e.max() < tol

    def rot(a):
        c, sn = dtype(np.cos(0.3)), dtype(np.sin(0.3))
        ax, ay = a[0].copy(), a[1].copy()
        a[0][:], a[1][:] = c * ax - sn * ay, sn * ax + c * ay

    t.update_particles_u(rot)
    ao = t.accs_o(0.01)
    for i in range(0, s, 997):
        ex = t.exact_acc_o(i).astype(np.float64)
        got = np.array([a[i] for a in ao], dtype=np.float64)
        assert abs(np.linalg.norm(ex) - np.linalg.norm(got)) / np.linalg.norm(ex) <= tol


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_narrow_code_trees_on_gpu(ndim, dtype):
    """tree<NDim, F, std::uint32_t, MAC> (32-bit Morton codes, the other UInt instantiation of the seam,
    rakau_rocm.cpp:333-370): shallow trees whose deepest-level leaves hold many particles. The header widens the node
    records at the seam; results against the 32-bit-code oracle on the identical tree."""
    rng = oracle.Rng(23)
    p = rng.uniform_particles(60000, 1.0, dtype, ndim=ndim)
    m, c = p[0], list(p[1:])
    z = c[2] if ndim == 3 else None
    for max_leaf_n, ncrit, theta in ((16, 128, 0.75), (2, 300, 0.5)):
        t = rakau_amd.Octree(c[0], c[1], z, m, max_leaf_n=max_leaf_n, ncrit=ncrit, code_bits=32)
        ot = oracle.Tree(c[0], c[1], z, m, max_leaf_n=max_leaf_n, ncrit=ncrit, ndim=ndim, code_bits=32)
        assert (t.n_nodes, t.n_crit) == (ot.n_nodes, ot.n_crit)
        got, ref = t.accs_pots_u(theta, eps=1e-3), ot.acc_pot(2, theta, eps=1e-3, nthreads=4)
        e = rel_err_vec(got, ref, ndim=ndim)
        tol, med = (5e-4, 1e-6) if dtype == np.float32 else (1e-11, 1e-14)
        assert e.max() <= tol and np.median(e) <= med
        assert rel_err(got[-1], ref[-1]).max() <= tol
        perm = t.perm().astype(np.int64)
        for u, o in zip(got, t.accs_pots_o(theta, eps=1e-3)):
            assert np.array_equal(o[perm], u)
