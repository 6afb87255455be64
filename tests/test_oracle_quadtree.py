"""Pins of the 2-dimensional (quadtree) flavour of the oracle. The reference's own tests hold little for quadtrees
(test/node_centre.cpp:23-53; test/coll.cpp uses them for the collision graph, out of scope), so the pins are: that
node-centre known-answer test, the 2-D Morton layout, and the reference's accuracy / G / ordering properties
(test/accuracy_acc.cpp, g_constant_acc.cpp, ordering_acc.cpp) restated with two coordinates."""
import numpy as np
import pytest

import oracle


def test_node_centre_quadtree():
    """test/node_centre.cpp:23-53: four unit masses at (+-1, +-1) in a box of 10 with max_leaf_n = 1 give 5 nodes whose
    geometric centres are the box centre and (+-box/4, +-box/4) in the order (-,-), (+,-), (-,+), (+,+). The centre is
    observable through bh_geom's delta = |COM - centre| and the children's nodal codes."""
    x = np.array([-1., -1., 1., 1.])
    y = np.array([-1., 1., -1., 1.])
    m = np.ones(4)
    t = oracle.Tree(x, y, None, m, box_size=10.0, max_leaf_n=1, mac="bh_geom", ndim=2)
    n = t.nodes()
    assert t.n_nodes == 5
    assert n["code"].tolist() == [1, 4, 5, 6, 7] and n["level"].tolist() == [0, 1, 1, 1, 1]
    assert np.array_equal(n["props"][1:, :2], np.array([[-1., -1.], [1., -1.], [-1., 1.], [1., 1.]]))
    assert np.allclose(n["props"][0], [0., 0., 4.])
    assert abs(n["dims"][0, 1]) < 1e-15  # root: COM = centre
    # children: centre (+-2.5, +-2.5), COM (+-1, +-1) -> delta = 1.5 * sqrt(2); dim = 5
    assert np.allclose(n["dims"][1:, 1], 1.5 * np.sqrt(2.0), rtol=0, atol=1e-14)
    assert np.array_equal(n["dims"][:, 0], [10., 5., 5., 5., 5.])
    one = oracle.Tree(np.array([-1.]), np.array([1.]), None, np.array([1.]), box_size=10.0, ndim=2)
    assert one.n_nodes == 1


def test_morton_2d_layout():
    """31 bits per coordinate, x in the even bits (tree_fwd.hpp:141-150, libmorton morton2D): the discretised
    coordinates of each particle are recovered from its code."""
    rng = np.random.default_rng(0)
    x, y = rng.uniform(-0.5, 0.5, 1000), rng.uniform(-0.5, 0.5, 1000)
    t = oracle.Tree(x, y, None, np.ones(1000), box_size=1.0, ndim=2)
    xs, ys, _ = t.parts_u()
    codes = t.codes_perms()["codes"]
    assert np.all(np.diff(codes.astype(np.int64)) >= 0) and int(codes.max()) < 2 ** 62

    def compact(c):
        out = np.zeros_like(c)
        for b in range(31):
            out |= ((c >> np.uint64(2 * b)) & np.uint64(1)) << np.uint64(b)
        return out

    disc = lambda v: np.floor((v / 1.0 + 0.5) * 2.0 ** 31).astype(np.uint64)
    assert np.array_equal(compact(codes), disc(xs)) and np.array_equal(compact(codes >> np.uint64(1)), disc(ys))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 2e-3)])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
def test_accuracy_and_properties_2d(dtype, tol, mac):
    rng = oracle.Rng(1)
    s = 2000
    m, x, y = rng.uniform_particles(s, 10.0, dtype, ndim=2)
    for max_leaf_n, ncrit in ((1, 1), (8, 16), (16, 128)):
        t = oracle.Tree(x, y, None, m, box_size=10.0, max_leaf_n=max_leaf_n, ncrit=ncrit, mac=mac, ndim=2)
        # theta -> 0: the tree sum is the direct sum (accuracy_acc.cpp:60-113 restated).
        acc = t.acc_pot(2, 0.001, nthreads=4)
        assert len(acc) == 3
        for i in range(0, s, 97):
            ex = t.exact(2, i).astype(np.float64)
            got = np.array([a[i] for a in acc], dtype=np.float64)
            assert abs(np.linalg.norm(ex[:2]) - np.linalg.norm(got[:2])) / np.linalg.norm(ex[:2]) <= tol
            assert abs(ex[2] - got[2]) / abs(ex[2]) <= tol
        # G scales exactly (g_constant_acc.cpp), ordered = scattered through perm (ordering_acc.cpp).
        a1, a2 = t.acc_pot(0, 0.75), t.acc_pot(0, 0.75, G=2.0)
        assert len(a1) == 2 and all(np.array_equal(2 * u, v) for u, v in zip(a1, a2))
        perm = t.codes_perms()["perm"].astype(np.int64)
        for u, o in zip(a1, t.acc_pot(0, 0.75, ordered=True)):
            assert np.array_equal(o[perm], u)
        # q = 1 and q = 2 agree with q = 0 bit for bit on the shared outputs.
        a3 = t.acc_pot(2, 0.75)
        assert all(np.array_equal(u, v) for u, v in zip(a1, a3[:2])) and np.array_equal(a3[2], t.acc_pot(1, 0.75)[0])
    assert np.array_equal(t.codes_perms()["inv_perm"][perm], np.arange(s, dtype=np.uint64))
