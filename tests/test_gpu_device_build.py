"""Device-side tree construction (rk_state_build) against the host builder / oracle: identical topology, codes,
permutation and critical nodes; node masses and centres of mass to rounding (sums over aligned runs of particles -- a
summation pyramid -- instead of the reference's serial particle sums); traversal results within the reference's tolerances."""
import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import rel_err_vec, rel_err, oracle_nodes_aos, state_from_oracle

pytestmark = pytest.mark.gpu


def compare(st, ot, x, y, z, m):
    dtype = x.dtype
    info = st.tree_info()
    assert info["device_built"] and info["box_size"] == ot.box_size
    assert (st.nparts, st.tree_size, st.n_crit) == (ot.nparts, ot.n_nodes, ot.n_crit)
    cp = ot.codes_perms()
    assert np.array_equal(st.download("codes"), cp["codes"])
    assert np.array_equal(st.download("perm"), cp["perm"])
    for k, ref in zip("xyzm", ot.parts_u()):
        assert np.array_equal(st.download(k), ref)
    assert np.array_equal(st.download("crit"), ot.crit_nodes())
    dn, on = st.download("nodes"), ot.nodes()
    for k in ("begin", "end", "n_children", "code", "level"):
        assert np.array_equal(dn[k], on[k]), k
    # Masses and centres of mass: rounding-level agreement (relative to the box for positions).
    eps = np.finfo(dtype).eps
    mass_d, mass_o = dn["props"][:, 3].astype(np.float64), on["props"][:, 3].astype(np.float64)
    assert np.max(np.abs(mass_d - mass_o) / np.maximum(mass_o, 1e-300)) < 64 * eps
    dpos = np.abs(dn["props"][:, :3].astype(np.float64) - on["props"][:, :3].astype(np.float64))
    assert dpos.max() < 256 * eps * ot.box_size
    key = "dim2" if ot.mac == "bh" else "dim"
    assert np.array_equal(dn[key], on["dims"][:, 0])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
def test_plummer_tree_and_traversal(dtype, mac):
    m, x, y, z = oracle.plummer(60000, dtype)
    ot = oracle.Tree(x, y, z, m, mac=mac)
    st = rakau_amd.State.build(x, y, z, m, mac=mac)
    compare(st, ot, x, y, z, m)
    mv = rakau_amd.mac_value_of(0.75, mac, dtype)
    got = st.acc_pot(2, mv)
    ref = ot.acc_pot(2, 0.75, nthreads=8)
    # MAC decisions can flip where a centre of mass moved by an ulp: bound = the reference's own (ordering_acc.cpp:93-97).
    tol = 2e-3 if dtype == np.float32 else 2e-11
    e = rel_err_vec(got, ref)
    assert e.max() <= tol and np.median(e) <= (1e-6 if dtype == np.float32 else 1e-14), (e.max(), np.median(e))
    assert rel_err(got[3], ref[3]).max() <= tol
    # Same thing in the original order through perm.
    perm = st.download("perm").astype(np.int64)
    ref_o = ot.acc_pot(2, 0.75, ordered=True, nthreads=8)
    # (per-particle vector norms for the accelerations: component-wise ratios blow up near zero components.)
    back = []
    for g in got:
        out = np.empty_like(g)
        out[perm] = g
        back.append(out)
    assert rel_err_vec(back, ref_o).max() <= tol
    assert rel_err(back[3], ref_o[3]).max() <= tol


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_parameter_matrix(dtype):
    rng = oracle.Rng(5)
    for s in (1, 2, 17, 300, 3000):
        m, x, y, z = rng.uniform_particles(s, 1.0, dtype)
        for max_leaf_n, ncrit in ((1, 1), (2, 16), (8, 128), (16, 256), (16, 1), (64, 64), (65, 300), (1000, 128)):
            for box in (1.0, None):
                ot = oracle.Tree(x, y, z, m, box_size=box or 0.0, max_leaf_n=max_leaf_n, ncrit=ncrit)
                st = rakau_amd.State.build(x, y, z, m, box_size=box, max_leaf_n=max_leaf_n, ncrit=ncrit)
                compare(st, ot, x, y, z, m)


def test_coincident_and_zero_mass_and_errors():
    rng = oracle.Rng(6)
    m, x, y, z = rng.uniform_particles(4000, 1.0, np.float64)
    x[:700], y[:700], z[:700] = 0.25, 0.25, -0.125  # 700 particles in one deepest-level cell
    m[100:200] = 0.0
    ot = oracle.Tree(x, y, z, m, box_size=1.0)
    st = rakau_amd.State.build(x, y, z, m, box_size=1.0)
    compare(st, ot, x, y, z, m)
    mv = rakau_amd.mac_value_of(0.6, "bh", np.float64)
    got = st.acc_pot(2, mv, eps2=1e-4)
    ref = ot.acc_pot(2, 0.6, eps=1e-2, nthreads=8)
    assert rel_err_vec(got, ref).max() < 1e-9
    mz = np.zeros_like(m)
    for mac in ("bh", "bh_geom"):
        st = rakau_amd.State.build(x, y, z, mz, box_size=1.0, mac=mac)
        compare(st, oracle.Tree(x, y, z, mz, box_size=1.0, mac=mac), x, y, z, mz)
        # Softened: 700 particles coincide, and 0 * inf would be NaN in the reference as well.
        for r in st.acc_pot(2, rakau_amd.mac_value_of(0.75, mac, np.float64), eps2=1e-4):
            assert np.all(r == 0)
    with pytest.raises(ValueError, match="outside the allowed bounds"):
        rakau_amd.State.build(x, y, z, m, box_size=0.4)
    with pytest.raises(ValueError, match="maximum number of particles per leaf must be nonzero"):
        rakau_amd.State.build(x, y, z, m, max_leaf_n=0)
    xb = x.copy()
    xb[5] = np.inf
    with pytest.raises(ValueError, match="non-finite"):
        rakau_amd.State.build(xb, y, z, m)


def test_non_finite_node_properties_are_reported():
    """A non-finite mass makes a centre of mass non-finite: the build refuses it with the reference's message (tree.hpp:1199-1204).
    The error bit is raised by the kernel that runs while the host already looks at the node counts, and comes back with the
    build's last look-up; the next build on the same thread is unaffected."""
    rng = oracle.Rng(7)
    for dtype in (np.float32, np.float64):
        m, x, y, z = rng.uniform_particles(5000, 1.0, dtype)
        mb = m.copy()
        mb[17] = np.inf
        for exact in (0, 1):
            rakau_amd.set_build_exact(exact)
            try:
                with pytest.raises(ValueError, match="centre of mass of a node produced a non-finite value"):
                    rakau_amd.State.build(x, y, z, mb, box_size=1.0)
                st = rakau_amd.State.build(x, y, z, m, box_size=1.0)
                assert st.nparts == 5000
            finally:
                rakau_amd.set_build_exact(0)


def test_build_time_4m():
    """Not a parity test: records the device build time next to the host build (printed with -s)."""
    import time
    from bench import plummer_numpy
    m, x, y, z = plummer_numpy(4_000_000, "float32")
    rakau_amd.State.build(x[:1000], y[:1000], z[:1000], m[:1000])  # warm up
    t0 = time.perf_counter()
    st = rakau_amd.State.build(x, y, z, m)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    t = rakau_amd.Octree(x, y, z, m)
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    hs = t.state()
    t_up = time.perf_counter() - t0
    print("\\n4M fp32: device build %.1f ms (incl. H2D of inputs); host build %.1f ms + state upload %.1f ms"
          % (t_dev * 1e3, t_host * 1e3, t_up * 1e3))
    assert (st.tree_size, st.n_crit) == (hs.tree_size, hs.n_crit)
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    a, b = st.acc_pot(0, mv), hs.acc_pot(0, mv)
    e = rel_err_vec(a, b)
    # The two trees differ by ulps in some centres of mass, which flips a handful of the ~1.5e8 MAC decisions; a flipped
    # decision moves the affected group by the Barnes-Hut truncation error of one node at theta = 0.75 (<~ 1e-2), as
    # between the reference's own scalar and SIMD node-property flavours. Everything else agrees to rounding.
    n_off = int((e > 1e-4).sum())
    print("device-built vs host-built tree: median %.2e, max %.2e, particles above 1e-4: %d" % (np.median(e), e.max(), n_off))
    # (fp32 coordinates of magnitude ~2000 carry ~1e-4 of absolute rounding in a centre of mass, whichever order the
    # particles are summed in: near-field monopoles move by ~1e-4 relative.)
    # Measured envelope of the DEFAULT (child -> parent) sums: max 2.7e-3, 515 particles above 1e-4; rk_set_build_exact(1)
    # removes the difference altogether (test_exact_build_4m_census_and_time). The bound is the envelope the reference's own
    # two builds show against each other on the same inputs: test_reference_simd_build_vs_scalar_build_envelope_4m below.
    assert np.median(e) < 1e-5 and e.max() < 5e-3 and n_off < 1000


def test_reference_simd_build_vs_scalar_build_envelope_4m():
    """The other half of the argument for the bound above. The reference's DEFAULT build sums a node's particles as
    batch_size interleaved partial sums, added horizontally, plus a scalar tail (tree.hpp:1134-1161); its scalar build
    (RAKAU_DISABLE_SIMD, the flavour the oracle restates and the host builder reproduces bit for bit) adds them one after the
    other (1162-1168). The two give centres of mass that differ by rounding, which flips a handful of the 1.5e8 MAC decisions
    of the 4M step -- exactly what the device builder's child -> parent sums do. Here the SAME 4M inputs are built with both
    associations (oracle.set_simd_width(8) = AVX2, 16 = AVX-512) and traversed on the GPU: the SIMD-flavoured trees sit in
    the same envelope against the scalar-flavoured one as the device-built tree does (max < 5e-3, fewer than 1000 of 4M
    particles above 1e-4, median at rounding level), so no bound tighter than that separates "built differently" from
    "built wrongly" -- and rk_set_build_exact(1) exists for callers who need the scalar flavour's bits."""
    from bench import plummer_numpy
    m, x, y, z = plummer_numpy(4_000_000, "float32")
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    res = {}
    try:
        for w in (1, 8, 16):
            oracle.set_simd_width(w)
            ot = oracle.Tree(x, y, z, m)
            st = state_from_oracle(ot)
            res[w] = (st.acc_pot(0, mv), ot.n_nodes, st.n_crit)
            del st, ot
    finally:
        oracle.set_simd_width(1)
    for w in (8, 16):
        assert res[w][1:] == res[1][1:]  # same topology, same critical nodes: only node properties differ
        e = rel_err_vec(res[w][0], res[1][0])
        n_off = int((e > 1e-4).sum())
        print("reference association, batch size %d vs scalar: median %.2e, max %.2e, particles above 1e-4: %d" % (w, np.median(e), e.max(), n_off))
        assert 0 < e.max() < 5e-3 and np.median(e) < 1e-5 and n_off < 1000


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_octree_device_builder_front_door(dtype):
    """kwargs::device_build through the C++ header (Octree(builder="device")): accessors, ordered outputs against the
    direct sum (test/ordering_acc.cpp:93-97 bounds) and update_particles_u (rebuild on the GPU) behave like the host
    builder's."""
    rng = oracle.Rng(2)
    s = 10000
    m, x, y, z = rng.uniform_particles(s, 1.0, dtype)
    td = rakau_amd.Octree(x, y, z, m, box_size=4.0, builder="device")
    th = rakau_amd.Octree(x, y, z, m, box_size=4.0)
    assert (td.n_nodes, td.n_crit, td.box_size) == (th.n_nodes, th.n_crit, th.box_size)
    for a, b in ((td.perm(), th.perm()), (td.inv_perm(), th.inv_perm()), (td.last_perm(), th.last_perm()),
                 (td.c_it_u(), th.c_it_u()), (td.crit_nodes(), th.crit_nodes())):
        assert np.array_equal(a, b)
    for a, b in zip(td.p_its_u(), th.p_its_u()):
        assert np.array_equal(a, b)
    nd, nh = td.nodes(), th.nodes()
    for k in ("begin", "end", "n_children", "code", "level"):
        assert np.array_equal(nd[k], nh[k])
    tol = 2e-3 if dtype == np.float32 else 2e-11

    def check(t):
        res = t.accs_o(0.01)
        for i in range(0, s, 997):
            ex = t.exact_acc_o(i).astype(np.float64)
            got = np.array([r[i] for r in res], dtype=np.float64)
            assert abs(np.linalg.norm(ex) - np.linalg.norm(got)) / np.linalg.norm(ex) <= tol

    check(td)
    c, sn = dtype(np.cos(0.7)), dtype(np.sin(0.7))

    def rot(a):
        ax, ay = a[0].copy(), a[1].copy()
        a[0][:], a[1][:] = c * ax - sn * ay, sn * ax + c * ay

    td.update_particles_u(rot)
    th.update_particles_u(rot)
    assert np.array_equal(td.perm(), th.perm()) and np.array_equal(td.last_perm(), th.last_perm())
    assert np.array_equal(td.inv_perm()[td.perm()], np.arange(s, dtype=np.uint64))
    check(td)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
def test_exact_build_is_bit_identical_to_the_host_builders(dtype, mac):
    """rk_set_build_exact(1): node sums in the reference's association (tree.hpp:1162-1168). The downloaded node records
    -- topology AND properties -- equal the oracle's byte for byte, so device-built and host-built states give the same
    results bit for bit (same MAC decisions, same interaction lists)."""
    rakau_amd.set_build_exact(True)
    try:
        for n in (3000, 70000):
            m, x, y, z = oracle.plummer(n, dtype)
            ot = oracle.Tree(x, y, z, m, mac=mac)
            sb = rakau_amd.State.build(x, y, z, m, mac=mac)
            assert sb.download("nodes").tobytes() == oracle_nodes_aos(ot).tobytes()
            hs = state_from_oracle(ot)
            mv = rakau_amd.mac_value_of(0.75, mac, dtype)
            for a, b in zip(sb.acc_pot(2, mv, eps2=1e-6), hs.acc_pot(2, mv, eps2=1e-6)):
                assert np.array_equal(a, b)
            assert sb.count_interactions(mv) == hs.count_interactions(mv)
        # Deep nests: two tight clumps (thousands of particles within 1e-4 of a point, one of them coincident) in a sparse
        # background -- long chains that start deep in the tree, heads with many nested first children, empty segments.
        rs = np.random.RandomState(3)
        nb, nc = 40000, 6000
        bg = rs.uniform(-1.0, 1.0, size=(nb, 3))
        c1 = np.array([0.31, -0.27, 0.11]) + 1e-4 * rs.standard_normal((nc, 3))
        c2 = np.tile(np.array([[-0.52, 0.44, -0.38]]), (nc // 4, 1))
        c2[nc // 8:] += 1e-6 * rs.standard_normal((nc // 4 - nc // 8, 3))
        pts = np.concatenate([bg, c1, c2]).astype(dtype)
        rs.shuffle(pts)
        mm = rs.uniform(0.5, 1.5, size=len(pts)).astype(dtype)
        xx, yy, zz = (np.ascontiguousarray(pts[:, k]) for k in range(3))
        ot = oracle.Tree(xx, yy, zz, mm, mac=mac, box_size=2.5)
        sb = rakau_amd.State.build(xx, yy, zz, mm, mac=mac, box_size=2.5)
        assert sb.download("nodes").tobytes() == oracle_nodes_aos(ot).tobytes()
        # Quadtree.
        rng = oracle.Rng(5)
        m, x, y = rng.uniform_particles(20000, 2.0, dtype, ndim=2)
        oq = oracle.Tree(x, y, None, m, mac=mac, ndim=2)
        sq = rakau_amd.State.build(x, y, None, m, mac=mac)
        assert sq.download("nodes").tobytes() == oracle_nodes_aos(oq).tobytes()
    finally:
        rakau_amd.set_build_exact(False)


def test_exact_build_4m_census_and_time():
    """BASELINE size: the exact device build reproduces the host builder's tree (identical interaction census and
    identical accelerations), and costs 25 ms, not the ~340 ms of a host build + upload."""
    import time
    from bench import plummer_numpy
    n = 4_000_000
    m, x, y, z = plummer_numpy(n, "float32")
    t = rakau_amd.Octree(x, y, z, m)
    hs = t.state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    rakau_amd.set_build_exact(True)
    try:
        rakau_amd.State.build(x[:2048], y[:2048], z[:2048], m[:2048]).close()
        t0 = time.perf_counter()
        sb = rakau_amd.State.build(x, y, z, m)
        dt_exact = time.perf_counter() - t0
    finally:
        rakau_amd.set_build_exact(False)
    t0 = time.perf_counter()
    sf = rakau_amd.State.build(x, y, z, m)
    dt_fast = time.perf_counter() - t0
    print("\\n4M device build from host arrays: exact %.1f ms, default %.1f ms" % (dt_exact * 1e3, dt_fast * 1e3))
    assert (sb.tree_size, sb.n_crit) == (hs.tree_size, hs.n_crit)
    assert sb.count_interactions(mv) == hs.count_interactions(mv)
    for a, b in zip(sb.acc_pot(0, mv), hs.acc_pot(0, mv)):
        assert np.array_equal(a, b)
    assert dt_exact < 0.08  # 24-27 ms measured (round 2: 60 ms): the root's chain of N dependent multiply-adds + 64 MB of H2D
    sf.close()


def class2_of(size):
    """rk_common.hpp class2_of_compute(): targets per lane R = c + 1 that minimises R / floor(64 / ceil(size / R))."""
    best, best_cost = -1, 0.0
    for c in range(4):
        tp = (size + c) // (c + 1)
        if tp > 64:
            continue
        cost = (c + 1) / (64 // tp)
        if best < 0 or cost < best_cost - 1e-12:
            best, best_cost = c, cost
    return best


@pytest.mark.parametrize("n,ncrit", [(3000, 128), (60000, 128), (150000, 256), (150000, 1300), (1500000, 128), (2300000, 128)])
def test_first_call_launch_order_made_on_the_device(n, ncrit):
    """Trees come with the launch order of their first call, made on the device with the tree (rk_build.hip bin_classes) -- for
    trees built on the device, for host trees converted there and for replicas.
    Up to 49152 critical nodes: eight queues, one per XCD region (eighths of the particle range), the critical nodes the wave kernels
    serve inside a queue by decreasing size (in steps of eight), ties in Morton order -- the heavy-first queues repeated calls get
    from the host; every such node exactly once, oversized nodes (ncrit = 1300) left to their own kernel.
    Beyond (up to 250000): the light-tail arrangement repeated calls get from the host -- per lane-mapping class and per XCD region
    (eight regions of equal particle count) one queue: the nodes of the region in Morton order, those below the first quartile of
    the class's sizes (estimated on a sample of at most 8192 nodes) at the end.
    Either way the first call gives the bits of a repeated call (RK_FIRST_ORDER=0 is one of the environments of
    tests/test_gpu_call_caches.py)."""
    import torch
    m, x, y, z = oracle.plummer(n, np.float32)
    ot = oracle.Tree(x, y, z, m, ncrit=ncrit, max_leaf_n=16 if ncrit < 1000 else 300)
    built = rakau_amd.State.build(x, y, z, m, ncrit=ncrit, max_leaf_n=ot.max_leaf_n)

    def fetch(st, what, count):
        ptr, nbytes = st.device_ptr(what)
        assert nbytes == 4 * count and ptr != 0, (what, nbytes, count)
        d_got = torch.zeros(count, dtype=torch.int32, device="cuda")
        rakau_amd._capi.check(rakau_amd._capi.lib().rk_device_memcpy(d_got.data_ptr(), ptr, nbytes, 0))
        torch.cuda.synchronize()
        return d_got.cpu().numpy().astype(np.int64)

    # (a replica makes the order from its own copy of the critical nodes: rk_state_clone / import / broadcast)
    for st in (built, state_from_oracle(ot), built.clone(0)):
        cr = st.crit_ranges()
        size = (cr[:, 1] - cr[:, 0]).astype(np.int64)
        wave = np.flatnonzero(size <= 256)
        if len(cr) <= 49152:
            # eight queues, one per XCD region; inside a queue sizes in steps of eight (those below sixteen together), ties in Morton order
            got = fetch(st, "first_order", len(wave))
            tab = fetch(st, "first_tab", 72)
            region = np.minimum(7, cr[wave, 0].astype(np.int64) * 8 // n)
            bucket = np.minimum(30, (256 - size[wave]) >> 3)
            pos = 0
            for xr in range(8):
                sel = wave[region == xr]
                expect = sel[np.lexsort((sel, bucket[region == xr]))]
                assert tab[xr] == pos and tab[8 + xr] == len(expect), (xr, tab[:16])
                assert np.array_equal(got[pos:pos + len(expect)], expect), xr
                pos += len(expect)
            assert pos == len(wave)
        else:
            got = fetch(st, "first_order", len(wave))
            tab = fetch(st, "first_tab", 72)
            cls = np.array([class2_of(int(v)) for v in size[wave]])
            region = np.minimum(7, cr[wave, 0].astype(np.int64) * 8 // n)
            pos = 0
            # the size from which a node of class c is bulk: the first quartile of the class's sizes among the first 8192 of every stride-th critical node
            stride = max(1, len(cr) // 8192)
            s_size = size[::stride][:8192]
            s_size = s_size[(s_size >= 1) & (s_size <= 256)]
            s_cls = np.array([class2_of(int(v)) for v in s_size])
            for c in range(4):
                of_class = np.sort(s_size[s_cls == c])
                thr = int(of_class[min(len(of_class) - 1, len(of_class) // 4)]) if len(of_class) else 0
                assert tab[64 + c] == thr
                for xr in range(8):
                    assert tab[c * 16 + xr] == pos
                    sel = wave[(cls == c) & (region == xr)]
                    expect = np.concatenate([sel[size[sel] >= thr], sel[size[sel] < thr]])
                    assert tab[c * 16 + 8 + xr] == len(expect)
                    assert np.array_equal(got[pos:pos + len(expect)], expect), (c, xr)
                    pos += len(expect)
            assert pos == len(wave)
        # and the first call, which runs over it, agrees with a repeated call (host plan) bit for bit
        mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
        a = st.acc_pot(0, mv)
        b = st.acc_pot(0, mv)
        c = st.acc_pot(0, mv)
        assert all(np.array_equal(u, v) and np.array_equal(u, w) for u, v, w in zip(a, b, c))


def test_first_call_light_tail_arrangement_fp64_and_quadtree():
    """The light-tail arrangement of first calls (trees of 49152-250000 critical nodes) through the fp64 kernels and the 2-D kernels:
    the first call (class kernels over the device-made queues, re-targeted executable graph) gives the bits of the repeated calls
    (host-made plan, captured graph), and the table tiles the wave-kernel nodes."""
    import torch
    for dtype, ndim, n in ((np.float64, 3, 2_300_000), (np.float32, 2, 2_300_000)):
        if ndim == 3:
            m, x, y, z = oracle.plummer(n, dtype)
            st = rakau_amd.State.build(x, y, z, m)
        else:
            m, x, y = oracle.Rng(7).uniform_particles(n, 3.0, dtype, ndim=2)
            st = rakau_amd.State.build(x, y, None, m)
        assert 49152 < st.n_crit <= 250000, st.n_crit
        ptr, nbytes = st.device_ptr("first_tab")
        assert nbytes == 4 * 72 and ptr != 0
        d_tab = torch.zeros(72, dtype=torch.int32, device="cuda")
        rakau_amd._capi.check(rakau_amd._capi.lib().rk_device_memcpy(d_tab.data_ptr(), ptr, nbytes, 0))
        torch.cuda.synchronize()
        tab = d_tab.cpu().numpy().astype(np.int64)
        cr = st.crit_ranges()
        n_wave = int(np.count_nonzero((cr[:, 1] - cr[:, 0]) <= 256))
        starts = np.array([tab[c * 16 + xr] for c in range(4) for xr in range(8)])
        lens = np.array([tab[c * 16 + 8 + xr] for c in range(4) for xr in range(8)])
        assert starts[0] == 0 and np.array_equal(starts[1:], np.cumsum(lens)[:-1]) and lens.sum() == n_wave
        mv = rakau_amd.mac_value_of(0.75, "bh", dtype)
        a = st.acc_pot(2, mv, eps2=1e-6)
        b = st.acc_pot(2, mv, eps2=1e-6)
        c = st.acc_pot(2, mv, eps2=1e-6)
        assert all(np.array_equal(u, v) and np.array_equal(u, w) for u, v, w in zip(a, b, c))
        st.close()
