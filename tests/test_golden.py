"""Committed golden vectors (tests/golden/oracle_cases.npz, see make_golden.py for provenance):
the oracle must still reproduce them bit for bit (CPU), and the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest

import oracle
import rakau_amd

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cases.npz"))
NAMES = sorted({k.split("/")[0] for k in G.files})


def _case(name):
    x, y, z, m = G[name + "/in"]
    theta, eps, Gc, mln, ncrit, n_nodes, n_crit, box = G[name + "/meta"]
    return (x, y, z, m), dict(theta=float(theta), eps=float(eps), G=float(Gc)), dict(
        max_leaf_n=int(mln), ncrit=int(ncrit), mac=str(G[name + "/mac"])), (int(n_nodes), int(n_crit), float(box))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    (x, y, z, m), kw, tkw, (n_nodes, n_crit, box) = _case(name)
    t = oracle.Tree(x, y, z, m, **tkw)
    assert (t.n_nodes, t.n_crit, t.box_size) == (n_nodes, n_crit, box)
    res = t.accs_pots_o(kw["theta"], eps=kw["eps"], G=kw["G"], nthreads=4)
    assert np.array_equal(np.stack(res), G[name + "/out"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_matches_golden(name):
    (x, y, z, m), kw, tkw, (n_nodes, n_crit, box) = _case(name)
    t = rakau_amd.Octree(x, y, z, m, **tkw)
    assert (t.n_nodes, t.n_crit, t.box_size) == (n_nodes, n_crit, box)
    got = np.stack(t.accs_pots_o(kw["theta"], eps=kw["eps"], G=kw["G"])).astype(np.float64)
    ref = G[name + "/out"].astype(np.float64)
    tol = 2e-5 if x.dtype == np.float32 else 1e-12
    err = np.linalg.norm(got[:3] - ref[:3], axis=0) / np.linalg.norm(ref[:3], axis=0)
    assert err.max() <= tol, err.max()
    assert np.max(np.abs(got[3] - ref[3]) / np.abs(ref[3])) <= tol


# ---- quadtrees and 32-bit codes (tests/golden/oracle_cases_nd.npz, make_golden_nd.py) ----------------------------
GN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cases_nd.npz"))
NAMES_ND = sorted({k.split("/")[0] for k in GN.files})


def _case_nd(name):
    arrs = GN[name + "/in"]
    ndim, bits, theta, eps, Gc, mln, ncrit, n_nodes, n_crit, box = GN[name + "/meta"]
    ndim = int(ndim)
    coords, m = list(arrs[:ndim]), arrs[ndim]
    return (coords, m, ndim, int(bits)), dict(theta=float(theta), eps=float(eps), G=float(Gc)), dict(
        max_leaf_n=int(mln), ncrit=int(ncrit), mac=str(GN[name + "/mac"])), (int(n_nodes), int(n_crit), float(box))


@pytest.mark.parametrize("name", NAMES_ND)
def test_oracle_reproduces_golden_nd(name):
    (c, m, ndim, bits), kw, tkw, (n_nodes, n_crit, box) = _case_nd(name)
    t = oracle.Tree(c[0], c[1], c[2] if ndim == 3 else None, m, ndim=ndim, code_bits=bits, **tkw)
    assert (t.n_nodes, t.n_crit, t.box_size) == (n_nodes, n_crit, box)
    res = t.accs_pots_o(kw["theta"], eps=kw["eps"], G=kw["G"], nthreads=4)
    assert np.array_equal(np.stack(res), GN[name + "/out"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES_ND)
def test_gpu_matches_golden_nd(name):
    (c, m, ndim, bits), kw, tkw, (n_nodes, n_crit, box) = _case_nd(name)
    t = rakau_amd.Octree(c[0], c[1], c[2] if ndim == 3 else None, m, code_bits=bits, **tkw)
    assert (t.n_nodes, t.n_crit, t.box_size) == (n_nodes, n_crit, box)
    got = np.stack(t.accs_pots_o(kw["theta"], eps=kw["eps"], G=kw["G"])).astype(np.float64)
    ref = GN[name + "/out"].astype(np.float64)
    tol = 5e-4 if m.dtype == np.float32 else 1e-11
    err = np.linalg.norm(got[:ndim] - ref[:ndim], axis=0) / np.linalg.norm(ref[:ndim], axis=0)
    assert err.max() <= tol and np.median(err) <= (1e-6 if m.dtype == np.float32 else 1e-14), (err.max(), np.median(err))
    assert np.max(np.abs(got[ndim] - ref[ndim]) / np.abs(ref[ndim])) <= tol
