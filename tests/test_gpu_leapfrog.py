"""Device-resident callers of the hot path: tree build from device pointers,
in-place rebuild, original-order outputs written by the kernel epilogue, and the kick-drift-kick loop of
benchmark/benchmark_leapfrog.cpp on top of them."""
import os
import sys

import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import rel_err_vec

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples"))


def torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


def same_tree(a, b, ndim=3):
    assert (a.nparts, a.tree_size, a.n_crit, a.max_group) == (b.nparts, b.tree_size, b.n_crit, b.max_group)
    assert a.tree_info() == b.tree_info()
    for what in ("x", "y", "z", "m", "codes", "perm", "crit") if ndim == 3 else ("x", "y", "m", "codes", "perm", "crit"):
        assert np.array_equal(a.download(what), b.download(what)), what
    na, nb = a.download("nodes"), b.download("nodes")
    assert na.tobytes() == nb.tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("q", [0, 1, 2])
def test_ordered_output_is_the_perm_scatter(dtype, q):
    """RK_OUT_ORDERED == Morton-order results scattered through perm (tree.hpp:3320-3330), bit for bit, for the full
    range and for a sub-range (which must leave the other elements alone)."""
    torch, dev = torch_dev()
    m, x, y, z = oracle.plummer(30000, dtype)
    st = rakau_amd.State.build(x, y, z, m)
    mv = rakau_amd.mac_value_of(0.75, "bh", dtype)
    perm = st.download("perm").astype(np.int64)
    tt = torch.float32 if dtype == np.float32 else torch.float64
    for variant in (0, 1):  # list kernel, depth-first kernel (+ the block kernel for big groups in both)
        st.set_variant(variant)
        ref = st.acc_pot(q, mv, eps2=1e-6)
        outs = [torch.full((st.nparts,), -7.0, dtype=tt, device=dev) for _ in ref]
        st.acc_pot_device(q, mv, [o.data_ptr() for o in outs], eps2=1e-6, ordered=True)
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            exp = np.empty_like(r)
            exp[perm] = r
            got = o.cpu().numpy()
            assert np.array_equal(got, exp), (variant, int((got != exp).sum()), np.nonzero(got[perm] != r)[0][:8],
                                              float(np.abs(got - exp).max()))
        # Sub-range aligned to critical nodes.
        cr = st.crit_ranges()
        b, e = int(cr[len(cr) // 3, 0]), int(cr[2 * len(cr) // 3, 0])
        outs = [torch.full((st.nparts,), -7.0, dtype=tt, device=dev) for _ in ref]
        st.acc_pot_device(q, mv, [o.data_ptr() for o in outs], eps2=1e-6, p_begin=b, p_end=e, ordered=True)
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            exp = np.full_like(r, -7.0)
            exp[perm[b:e]] = r[b:e]
            assert np.array_equal(o.cpu().numpy(), exp)
    st.set_variant(0)


def test_ordered_output_needs_perm_and_set_perm_provides_it():
    torch, dev = torch_dev()
    rng = oracle.Rng(3)
    m, x, y, z = rng.uniform_particles(5000, 1.0, np.float64)
    t = rakau_amd.Octree(x, y, z, m, box_size=1.0)
    st = t.state()
    mv = rakau_amd.mac_value_of(0.5, "bh", np.float64)
    outs = [torch.zeros(5000, dtype=torch.float64, device=dev) for _ in range(3)]
    with pytest.raises(ValueError, match="needs the permutation"):
        st.acc_pot_device(0, mv, [o.data_ptr() for o in outs], ordered=True)
    with pytest.raises(ValueError, match="invalid permutation entry"):
        st.set_perm(np.full(5000, 5000, dtype=np.uint64))
    st.set_perm(t.perm())
    st.acc_pot_device(0, mv, [o.data_ptr() for o in outs], ordered=True)
    torch.cuda.synchronize()
    ref = t.accs_o(0.5)
    for o, r in zip(outs, ref):
        assert np.array_equal(o.cpu().numpy(), r)
    # The host entry point scatters on the device as well (whole range only); unknown flag bits are refused.
    from rakau_amd import _capi
    import ctypes as C
    hout = [np.zeros(5000) for _ in range(3)]
    ptrs = (C.c_void_p * 4)(*[o.ctypes.data for o in hout], None)
    _capi.check(_capi.lib().rk_acc_pot(st._h, 0, 0, 5000, ptrs, mv, 1.0, 0.0, _capi.RK_OUT_ORDERED))
    for o, r in zip(hout, ref):
        assert np.array_equal(o, r)
    with pytest.raises(ValueError, match="invalid output flags"):
        _capi.check(_capi.lib().rk_acc_pot(st._h, 0, 0, 5000, ptrs, mv, 1.0, 0.0, 4))
    with pytest.raises(ValueError, match="takes the whole range"):
        _capi.check(_capi.lib().rk_acc_pot(st._h, 0, 0, int(st.crit_ranges()[1, 0]), ptrs, mv, 1.0, 0.0, _capi.RK_OUT_ORDERED))


def test_set_perm_between_identical_ordered_calls():
    """Two identical ordered calls replay a captured launch graph; a set_perm() in between must take effect (the
    graph key covers the permutation, and set_perm drops the captured graph), also for replicas made by
    rk_state_export/import, which carry the permutation."""
    torch, dev = torch_dev()
    m, x, y, z = oracle.plummer(20000, np.float32)
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    perm = t.perm()
    st.set_perm(perm)
    outs = [torch.zeros(20000, dtype=torch.float32, device=dev) for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(3):  # direct launch, capture, replay
        st.acc_pot_device(0, mv, ptrs, ordered=True)
    torch.cuda.synchronize()
    first = [o.cpu().numpy().copy() for o in outs]
    ref = t.accs_o(0.75)
    for g, r in zip(first, ref):
        assert np.array_equal(g, r)
    # A different permutation (reversed original order), same outputs, same parameters.
    perm2 = (np.uint64(19999) - perm).astype(np.uint64)
    st.set_perm(perm2)
    for _ in range(3):
        st.acc_pot_device(0, mv, ptrs, ordered=True)
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        assert np.array_equal(o.cpu().numpy(), r[::-1])
    # Replica: ordered output works and follows the exported permutation.
    eptrs, nbytes, meta = st.export()
    clone = rakau_amd.State.from_buffers(0, eptrs, nbytes, meta)
    outs2 = [torch.zeros(20000, dtype=torch.float32, device=dev) for _ in range(3)]
    clone.acc_pot_device(0, mv, [o.data_ptr() for o in outs2], ordered=True)
    torch.cuda.synchronize()
    for o, r in zip(outs2, ref):
        assert np.array_equal(o.cpu().numpy(), r[::-1])
    # A meta block that disagrees with the buffers is rejected (no out-of-bounds device reads).
    bad = list(meta)
    bad[3] += 64
    with pytest.raises(ValueError, match="does not match the meta block"):
        rakau_amd.State.from_buffers(0, eptrs, nbytes, bad)
    clone.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_build_from_device_pointers_and_rebuild(dtype):
    """rk_state_build_device == rk_state_build on the same particles; rk_state_rebuild_device == a fresh build, through
    growing and shrinking particle counts (recycled buffers must not leak stale data into the tree)."""
    torch, dev = torch_dev()
    tt = torch.float32 if dtype == np.float32 else torch.float64
    rng = oracle.Rng(11)

    def dev_parts(n, scale):
        m, x, y, z = rng.uniform_particles(n, scale, dtype)
        ts = [torch.as_tensor(v).to(dev) for v in (x, y, z, m)]
        return (x, y, z, m), ts

    host, ts = dev_parts(40000, 1.0)
    torch.cuda.synchronize()
    st = rakau_amd.State.build_device([t.data_ptr() for t in ts], 40000, dtype, mac="bh_geom")
    same_tree(st, rakau_amd.State.build(*host, mac="bh_geom"))
    assert st.device_ptr("parts")[1] == 40000 * 4 * np.dtype(dtype).itemsize
    assert st.device_ptr("perm")[1] == 40000 * 4 and st.device_ptr("codes")[1] == 40000 * 8
    mv = rakau_amd.mac_value_of(0.75, "bh_geom", dtype)
    for n, scale, box in ((40000, 3.0, None), (90000, 1.0, 2.0), (1000, 1.0, None), (17, 1.0, None), (40000, 1.0, 4.0)):
        host, ts = dev_parts(n, scale)
        torch.cuda.synchronize()
        st.rebuild_device([t.data_ptr() for t in ts], nparts=n, box_size=box)
        fresh = rakau_amd.State.build(*host, mac="bh_geom", box_size=box)
        same_tree(st, fresh)
        for a, b in zip(st.acc_pot(2, mv), fresh.acc_pot(2, mv)):
            assert np.array_equal(a, b)
    # A failed rebuild leaves an empty, usable state.
    bad = ts[0].clone()
    bad[5] = float("inf")
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="non-finite"):
        st.rebuild_device([bad.data_ptr()] + [t.data_ptr() for t in ts[1:]], nparts=40000)
    assert st.nparts == 0 and st.n_crit == 0
    st.rebuild_device([t.data_ptr() for t in ts], nparts=40000, box_size=4.0)
    same_tree(st, fresh)
    rakau_amd._capi.lib().rk_pool_trim()
    st.rebuild_device([t.data_ptr() for t in ts], nparts=40000, box_size=4.0)
    same_tree(st, fresh)


PARTIAL_SORT_CODE = r"""
import sys
import numpy as np, torch
import oracle, rakau_amd
sys.path.insert(0, "tests")
from test_gpu_leapfrog import same_tree
dev = torch.device("cuda", 0)
for dtype, nd in ((np.float32, 3), (np.float64, 3), (np.float32, 2)):
    m, x, y, z = oracle.plummer(50000, dtype)
    rs = np.random.RandomState(3)
    pos = [x.copy(), y.copy()] + ([z.copy()] if nd == 3 else [])
    tm = torch.as_tensor(m).to(dev)
    ts = [torch.as_tensor(v).to(dev) for v in pos]
    torch.cuda.synchronize()
    st = rakau_amd.State.build_device([t.data_ptr() for t in ts] + [tm.data_ptr()], 50000, dtype)
    for step in range(6):
        # small moves: the tree keeps its depth (partial sort, no second try); step 4 squeezes everything into a corner of the box
        # (a deeper tree: the first try must be found wanting and repeated with all bits)
        for k in range(nd):
            pos[k] = (pos[k] * (0.01 if step == 4 else 1.0) + dtype(1e-3) * rs.standard_normal(50000).astype(dtype)).astype(dtype)
        ts = [torch.as_tensor(v).to(dev) for v in pos]
        torch.cuda.synchronize()
        box = 60.0 if step >= 4 else None
        st.rebuild_device([t.data_ptr() for t in ts] + [tm.data_ptr()], nparts=50000, box_size=box)
        if nd == 3:
            fresh = rakau_amd.State.build(pos[0], pos[1], pos[2], m, box_size=box)
        else:
            fresh = rakau_amd.State.build(pos[0], pos[1], None, m, box_size=box)
        same_tree(st, fresh, nd)
print("PARTIAL_SORT_OK")
"""


@pytest.mark.parametrize("bias", ["1", "2", "-3", "-100"])
def test_rebuild_with_a_partial_sort_equals_a_fresh_build(bias):
    """A rebuild sorts only the code bits of the levels the previous tree used (+ bias) and orders the insides of the leaves itself
    (rk_build.hip k_local_sort); when the new tree turns out deeper, the front of the build runs again with all bits. Either way the
    state must equal a fresh build bit for bit. RK_SORT_MIN=0 puts these 50k-particle builds on the onesweep path, where the
    partial sort lives (default: from 2^20 particles); bias -3 makes every first try fail, -100 switches the partial sort off."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, RK_SORT_MIN="0", RK_SORT_PARTIAL=bias, RK_SORT_TRACE="1")
    env["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "tests"), env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", PARTIAL_SORT_CODE], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0 and "PARTIAL_SORT_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stderr.splitlines() if l.startswith("rk_build:")]
    import re
    partial = [l for l in lines if (lambda mm: mm and int(mm.group(1)) < int(mm.group(2)))(re.search(r"sorted levels (\d+) of (\d+)", l))]
    again = [l for l in lines if "again with all bits" in l]
    if bias == "-100":
        assert not partial and not again
    elif bias == "-3":
        assert again
    else:
        # the small moves keep the depth: partial sorts without a second try; the squeeze needs one
        assert len(partial) > len(again) >= 1, lines


def test_rebuild_on_the_public_library_sort_equals_a_fresh_build():
    """RK_SORT_MIN=-1 is the path a build against another rocPRIM than 4.2.0 takes (rk_build.hip RK_ONESWEEP_INTERNALS): every sort
    through hipcub::DeviceRadixSort, no partial-key rebuilds. Same sequence of rebuilds as the partial-sort test, same comparison
    with fresh host builds bit for bit."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, RK_SORT_MIN="-1", RK_SORT_TRACE="1")
    env.pop("RK_SORT_PARTIAL", None)
    env["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "tests"), env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", PARTIAL_SORT_CODE], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0 and "PARTIAL_SORT_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    import re
    lines = [l for l in out.stderr.splitlines() if l.startswith("rk_build:")]
    partial = [l for l in lines if (lambda mm: mm and int(mm.group(1)) < int(mm.group(2)))(re.search(r"sorted levels (\d+) of (\d+)", l))]
    assert not partial, partial


SORT_PATHS_CODE = r"""
import hashlib, sys
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy
dev = torch.device("cuda", 0)
n = 2_000_000
m, x, y, z = plummer_numpy(n, "float32")
h = hashlib.sha256()
tm = torch.as_tensor(m).to(dev)
ts = [torch.as_tensor(v).to(dev) for v in (x, y, z)]
torch.cuda.synchronize()
st = rakau_amd.State.build_device([t.data_ptr() for t in ts] + [tm.data_ptr()], n, np.float32)
rs = np.random.RandomState(11)
for step in range(3):
    for what in ("codes", "perm", "crit", "x", "m"):
        h.update(st.download(what).tobytes())
    h.update(st.download("nodes").tobytes())
    for r in st.acc_pot(0, rakau_amd.mac_value_of(0.75, "bh", np.float32)):
        h.update(r.tobytes())
    x = (x + np.float32(1e-3) * rs.standard_normal(n).astype(np.float32)).astype(np.float32)
    ts[0] = torch.as_tensor(x).to(dev)
    torch.cuda.synchronize()
    st.rebuild_device([t.data_ptr() for t in ts] + [tm.data_ptr()], nparts=n)
print("SORT_PATHS_HASH", h.hexdigest(), st.tree_size, st.n_crit)
"""


def test_2m_rebuild_onesweep_and_library_sorts_give_the_same_tree():
    """2M particles (above 2^20: the default build sorts with the onesweep launch sequence on rocPRIM's internals, 9-bit
    digits, partial keys in the rebuilds) against the public-API path (RK_SORT_MIN=-1): codes, permutation, critical nodes, node
    records and the accelerations of a build and two rebuilds hash to the same value."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    got = {}
    for name, extra in (("onesweep", {}), ("library", {"RK_SORT_MIN": "-1"})):
        env = dict(os.environ, **extra)
        if not extra:
            env.pop("RK_SORT_MIN", None)
        env["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "tests"), env.get("PYTHONPATH", "")])
        out = subprocess.run([sys.executable, "-c", SORT_PATHS_CODE], capture_output=True, text=True, timeout=900, env=env, cwd=root)
        line = [l for l in out.stdout.splitlines() if l.startswith("SORT_PATHS_HASH")]
        assert out.returncode == 0 and line, out.stdout[-2000:] + out.stderr[-4000:]
        got[name] = line[0]
    assert got["onesweep"] == got["library"], got


def cpu_leapfrog(x, y, z, vx, vy, vz, m, dt, steps, theta, eps):
    """The same KDK loop with the oracle as force engine (float64 bookkeeping of the same operations)."""
    pos = [x.copy(), y.copy(), z.copy()]
    vel = [vx.copy(), vy.copy(), vz.copy()]
    dtype = x.dtype
    h = dtype.type(0.5 * dt)
    dtt = dtype.type(dt)

    def accs():
        return oracle.Tree(pos[0], pos[1], pos[2], m).acc_pot(0, theta, eps=eps, ordered=True, nthreads=8)

    a = accs()
    for _ in range(steps):
        for k in range(3):
            vel[k] += a[k] * h
            pos[k] += vel[k] * dtt
        a = accs()
        for k in range(3):
            vel[k] += a[k] * h
    return pos, vel


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_leapfrog_tracks_the_oracle_driven_integration(dtype):
    import leapfrog
    n0, dt, steps, theta = 20000, 1e-3, 5, 0.75
    x, y, z, vx, vy, vz = leapfrog.plummer_with_velocities(n0, seed=4, dtype=dtype)
    n = x.size
    m = np.full(n, 1.0 / n, dtype=dtype)
    eps = 0.45 * n ** -0.73
    lf = leapfrog.Leapfrog(x, y, z, vx, vy, vz, m, dt, theta, eps)
    for _ in range(steps):
        lf.step()
    pos, vel = cpu_leapfrog(x, y, z, vx, vy, vz, m, dt, steps, theta, eps)
    # Displacements over the run are ~ v * steps * dt ~ 5e-3; compare them (not the positions) so that the check is
    # sensitive: the two engines agree to the traversal tolerance of the reference's tests (ordering_acc.cpp:93-97).
    tol = 2e-3 if dtype == np.float32 else 1e-9
    d_gpu = [p.cpu().numpy().astype(np.float64) - p0 for p, p0 in zip(lf.pos, (x, y, z))]
    d_cpu = [p.astype(np.float64) - p0 for p, p0 in zip(pos, (x, y, z))]
    assert np.median(rel_err_vec(d_gpu, d_cpu)) < (1e-4 if dtype == np.float32 else 1e-12)
    assert np.percentile(rel_err_vec(d_gpu, d_cpu), 99.9) < tol
    v_gpu = [v.cpu().numpy() for v in lf.vel]
    assert np.percentile(rel_err_vec(v_gpu, vel), 99.9) < tol


def test_leapfrog_conserves_energy_and_reports():
    import leapfrog
    res = leapfrog.run(nparts=100000, steps=20, warmup=0, timestep=1e-3, track_integrals=True)
    assert res["nparts"] > 97000 and res["value"] > 0
    # Plummer model in virial equilibrium: 2K/|W| = 1 up to sampling noise; KDK at dt = 1e-3 over 20 steps conserves
    # the total energy far better than 1e-3.
    assert abs(res["virial_2K_over_W"] - 1.0) < 0.05
    assert res["energy_rel_drift"] < 1e-3
    assert all(abs(c) < 0.05 for c in res["com_end"])
    print("\nleapfrog 100k:", {k: res[k] for k in ("value", "ms_per_step", "ms_rebuild", "ms_traversal", "energy_rel_drift")})


def test_native_leapfrog_harness():
    """examples/leapfrog.hip (C ABI only, own HIP integrator kernels): builds, runs, conserves energy, and reports the
    same tree statistics as the Python harness for the same particle count."""
    import json
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(root, "examples", "leapfrog"), "--nparts", "100000", "--steps", "20", "--warmup", "0",
                          "--timestep", "1e-3", "--track-integrals"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert 97000 < res["nparts"] <= 100000 and res["value"] > 0
    assert abs(res["virial_2K_over_W"] - 1.0) < 0.05 and res["energy_rel_drift"] < 1e-3
    assert res["n_crit"] > 1000 and res["tree_size"] > res["n_crit"]
    print("\nnative leapfrog 100k:", {k: res[k] for k in ("value", "ms_per_step", "ms_rebuild", "ms_traversal", "energy_rel_drift")})


def test_new_entry_points_argument_checks():
    """Error behaviour of the device-resident entry points: status codes and messages, no crashes."""
    import ctypes as C
    from rakau_amd import _capi
    lib = _capi.lib()
    m, x, y, z = oracle.plummer(5000, np.float32)
    st = rakau_amd.State.build(x, y, z, m)
    h = C.c_void_p()
    parts = (C.c_void_p * 4)(x.ctypes.data, y.ctypes.data, z.ctypes.data, m.ctypes.data)
    for ndim in (1, 4):
        with pytest.raises(ValueError, match="ndim must be 2"):
            _capi.check(lib.rk_state_build_nd(C.byref(h), ndim, 0, 0, 0, parts, 0, 5000, 0.0, 16, 128))
    with pytest.raises(ValueError, match="null particle array"):
        _capi.check(lib.rk_state_build_nd(C.byref(h), 3, 0, 0, 0, (C.c_void_p * 4)(x.ctypes.data, None, None, None), 0, 5000,
                                          0.0, 16, 128))
    with pytest.raises(ValueError, match="critical number of particles"):
        _capi.check(lib.rk_state_build_nd(C.byref(h), 3, 0, 0, 0, parts, 0, 5000, 0.0, 16, 0))
    with pytest.raises(ValueError, match="box size must be a finite non-negative"):
        _capi.check(lib.rk_state_build_nd(C.byref(h), 3, 0, 0, 0, parts, 0, 5000, -1.0, 16, 128))
    ptr, nb = C.c_void_p(), C.c_int64()
    with pytest.raises(ValueError, match="invalid selector"):
        _capi.check(lib.rk_state_device_ptr(st._h, 9, C.byref(ptr), C.byref(nb)))
    with pytest.raises(ArithmeticError, match="finite and positive"):
        st.group_work(-1.0)
    w = st.group_work(rakau_amd.mac_value_of(0.75, "bh", np.float32))
    cen = st.count_interactions(rakau_amd.mac_value_of(0.75, "bh", np.float32))
    assert w.shape == (st.n_crit,) and int(w.sum()) == cen["com"] + cen["pp"] + cen["self"]
    # A sub-range call on a device-built state (lazy host mirrors) and the full call agree bit for bit.
    cr = st.crit_ranges()
    cut = int(cr[len(cr) // 2, 0])
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    full = st.acc_pot(0, mv)
    lo, hi = st.acc_pot(0, mv, p_begin=0, p_end=cut, offset_output=False), st.acc_pot(0, mv, p_begin=cut, offset_output=False)
    for f, a, b in zip(full, lo, hi):
        assert np.array_equal(f, np.concatenate([a, b]))
    big = cr[np.argmax(cr[:, 1] - cr[:, 0] > 1)]
    with pytest.raises(ValueError, match="critical node boundaries"):
        st.acc_pot(0, mv, p_begin=int(big[0]) + 1, p_end=st.nparts)
    # Empty input: an empty, usable state.
    e = np.zeros(0, dtype=np.float32)
    empty = rakau_amd.State.build(e, e, e, e)
    assert (empty.nparts, empty.tree_size, empty.n_crit) == (0, 0, 0) and empty.crit_ranges().shape == (0, 2)
