"""The CPU engine of the C++ header (include/rakau_amd/cpu_engine.hpp): the engine that runs the host share of
kwargs::split (tree.hpp:3047-3113 of the reference) and bench.py's cpu_baseline. No GPU involved.

 * scalar flavour: the arithmetic and summation order of the reference's scalar branch -> equal to the oracle bit for bit;
 * SIMD flavours: same interaction lists, same per-target order outside the critical node; sqrt + divide ('simd_exact')
   or rsqrt + one Newton step for fp32 ('auto', the reference's AVX fast path, detail/simd.hpp:76-146) -> rounding level;
 * split = {1} (the reference's "CPU only") goes through the public acc/pot surface; the default split needs the GPU and
   fails loudly without one."""
import os

import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import rel_err, rel_err_vec


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def no_gpu():
    return rakau_amd._capi.lib().rk_has_accelerator() == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mac", ["bh", "bh_geom"])
def test_scalar_flavour_equals_the_oracle_bit_for_bit(dtype, mac):
    m, x, y, z = oracle.plummer(12000, dtype)
    t = rakau_amd.Octree(x, y, z, m, mac=mac)
    ot = oracle.Tree(x, y, z, m, mac=mac)
    for q, eps, G in ((0, 0.0, 1.0), (1, 1e-3, 1.0), (2, 1e-3, 1.5)):
        ref = ot.acc_pot(q, 0.75, eps=eps, G=G, nthreads=4)
        got = t.cpu_acc_pot_u(q, 0.75, eps=eps, G=G, flavour="scalar", nthreads=3)
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("flavour", ["simd_exact", "auto"])
def test_simd_flavours_agree_to_rounding(dtype, flavour):
    m, x, y, z = oracle.plummer(30000, dtype)
    t = rakau_amd.Octree(x, y, z, m)
    ot = oracle.Tree(x, y, z, m)
    for eps in (0.0, 1e-3):
        ref = ot.acc_pot(2, 0.75, eps=eps, nthreads=4)
        got = t.cpu_acc_pot_u(2, 0.75, eps=eps, flavour=flavour)
        e = rel_err_vec(got, ref)
        pe = rel_err(got[3], ref[3])
        # fp32: 24-bit arithmetic over ~1500 terms; the rsqrt + Newton flavour carries ~22-23 good bits per term.
        tol_max, tol_med = (2e-5, 1e-6) if dtype == np.float32 else (1e-12, 1e-14)
        assert e.max() < tol_max and np.median(e) < tol_med, (e.max(), np.median(e))
        assert pe.max() < tol_max
    # Deterministic whatever the number of threads (critical nodes are independent).
    a = t.cpu_acc_pot_u(0, 0.75, flavour=flavour, nthreads=1)
    b = t.cpu_acc_pot_u(0, 0.75, flavour=flavour, nthreads=5)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def test_quadtree_narrow_codes_and_reference_matrix():
    """2-D trees, 32-bit codes, and the reference's size x max_leaf_n x ncrit matrix (test/accuracy_acc.cpp:41-44)
    incl. critical nodes far larger than a SIMD batch, at theta = 0.001 (everything opened) and 0.75."""
    rng = oracle.Rng(7)
    m, x, y = rng.uniform_particles(3000, 4.0, np.float64, ndim=2)
    tq = rakau_amd.Quadtree(x, y, m, box_size=4.0)
    oq = oracle.Tree(x, y, None, m, box_size=4.0, ndim=2)
    for fl in ("scalar", "simd_exact"):
        got = tq.cpu_acc_pot_u(2, 0.5, eps=0.01, flavour=fl)
        ref = oq.acc_pot(2, 0.5, eps=0.01)
        assert len(got) == 3
        assert rel_err_vec(got, ref, ndim=2).max() < 1e-12 and rel_err(got[2], ref[2]).max() < 1e-12
    m, x, y, z = rng.uniform_particles(2000, 1.0, np.float32)
    tn = rakau_amd.Octree(x, y, z, m, box_size=1.0, code_bits=32)
    on = oracle.Tree(x, y, z, m, box_size=1.0, code_bits=32)
    for g, r in zip(tn.cpu_acc_pot_u(0, 0.6, flavour="scalar"), on.acc_pot(0, 0.6)):
        assert np.array_equal(g, r)
    for s in (10, 100, 1000):
        for mln in (1, 8, 16):
            for ncrit in (1, 16, 128, 256):
                m, x, y, z = rng.uniform_particles(s, 1.0, np.float64)
                t = rakau_amd.Octree(x, y, z, m, box_size=1.0, max_leaf_n=mln, ncrit=ncrit)
                ot = oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=mln, ncrit=ncrit)
                for theta in (0.001, 0.75):
                    ref = ot.acc_pot(2, theta)
                    for g, r in zip(t.cpu_acc_pot_u(2, theta, flavour="scalar"), ref):
                        assert np.array_equal(g, r)
                    got = t.cpu_acc_pot_u(2, theta, flavour="simd_exact")
                    assert rel_err_vec(got, ref).max() < 1e-11 and rel_err(got[3], ref[3]).max() < 1e-11


def test_properties_of_the_reference_tests():
    """G scaling is exact (test/g_constant_acc.cpp:66-88), zero masses give exact zeros (test/zero_masses.cpp:54-75),
    coincident particles stay finite with softening (test/softening_acc.cpp)."""
    rng = oracle.Rng(9)
    m, x, y, z = rng.uniform_particles(4000, 1.0, np.float32)
    t = rakau_amd.Octree(x, y, z, m, box_size=1.0)
    a, b, zero = t.cpu_acc_pot_u(2, 0.75), t.cpu_acc_pot_u(2, 0.75, G=2.0), t.cpu_acc_pot_u(2, 0.75, G=0.0)
    for u, v, w in zip(a, b, zero):
        assert np.array_equal(u * np.float32(2), v) and not w.any()
    tz = rakau_amd.Octree(x, y, z, np.zeros_like(m), box_size=1.0)
    for v in tz.cpu_acc_pot_u(2, 0.75):
        assert not v.any()
    x2, y2, z2 = x.copy(), y.copy(), z.copy()
    x2[:300], y2[:300], z2[:300] = x2[0], y2[0], z2[0]
    tc = rakau_amd.Octree(x2, y2, z2, m, box_size=1.0)
    for fl in ("scalar", "auto"):
        for v in tc.cpu_acc_pot_u(2, 0.75, eps=0.1, flavour=fl):
            assert np.all(np.isfinite(v))
    with pytest.raises(ArithmeticError, match="MAC value must be finite and positive"):
        t.cpu_acc_pot_u(0, -1.0)
    with pytest.raises(ArithmeticError, match="softening length must be finite and non-negative"):
        t.cpu_acc_pot_u(0, 0.75, eps=-1.0)


def test_public_surface_split_one_is_the_cpu_engine():
    """split = {x} is the reference's "CPU only" (tree.hpp:3239-3242): the call needs no GPU, _u and _o agree through
    perm, and the result is the engine's. Without a GPU the default (device) path fails loudly."""
    m, x, y, z = oracle.plummer(8000, np.float32)
    t = rakau_amd.Octree(x, y, z, m)
    eng = t.cpu_acc_pot_u(2, 0.75, eps=1e-3, G=3.0)
    u = t.accs_pots_u(0.75, eps=1e-3, G=3.0, split=[1.0])
    o = t.accs_pots_o(0.75, eps=1e-3, G=3.0, split=[0.7])
    perm = t.perm().astype(np.int64)
    for e, a, b in zip(eng, u, o):
        assert np.array_equal(e, a) and np.array_equal(b[perm], a)
    assert np.array_equal(t.pots_u(0.75, split=[1.0]), t.cpu_acc_pot_u(1, 0.75)[0])
    for bad, msg in (((float("inf"),), "cannot contain non-finite"), ((-1.0,), "only non-negative"), ((0.0,), "cannot all be zero")):
        with pytest.raises(ValueError, match=msg):
            t.accs_u(0.75, split=bad)
    if no_gpu():
        with pytest.raises(RuntimeError, match="no gfx950 accelerator"):
            t.accs_u(0.75)
        with pytest.raises((RuntimeError, ValueError)):
            t.accs_u(0.75, split=[0.5, 0.5])


def test_avx512_flavour_equals_the_avx2_flavour_bit_for_bit():
    """On a CPU with AVX-512 the library runs the engine's AVX-512 build (librakau_amd_cpu512.so, rk_cpu_engine_run). A
    lane is a target whatever the batch width, so with sqrt + divide arithmetic the two builds must agree bit for bit;
    with RAKAU_AMD_CPU_ISA=avx2 (read once per process, hence the subprocess) the hand-off is disabled."""
    import subprocess
    import sys
    import tempfile
    script = (
        "import sys, numpy as np; sys.path.insert(0, %r); import oracle, rakau_amd\n"
        "m, x, y, z = oracle.plummer(20000, np.float64)\n"
        "t = rakau_amd.Octree(x, y, z, m)\n"
        "np.save(sys.argv[1], np.stack(t.cpu_acc_pot_u(2, 0.6, eps=1e-3, flavour='simd_exact')))\n" % ROOT)
    res = {}
    for isa in ("avx512", "avx2"):
        with tempfile.NamedTemporaryFile(suffix=".npy") as f:
            env = dict(os.environ, RAKAU_AMD_CPU_ISA=isa)
            subprocess.run([sys.executable, "-c", script, f.name], check=True, env=env, timeout=300)
            res[isa] = np.load(f.name)
    assert np.array_equal(res["avx512"], res["avx2"])
