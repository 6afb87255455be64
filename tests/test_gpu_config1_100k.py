"""BASELINE configuration 1 at its own size: 3D fp32, 100k-particle Plummer sphere, theta = 0.75, accs_u() -- the reference's
CPU-runnable case (benchmark/benchmark_acc.cpp with the README's parameters at N = 1e5). Both engines behind the front
door against the oracle on the WHOLE problem: the header's CPU engine (what a call without a GPU share runs) and the HIP
path on all of the tree's critical nodes, every kernel variant."""
import numpy as np
import pytest

import oracle
import rakau_amd
from bench import plummer_numpy
from helpers import rel_err_vec

pytestmark = pytest.mark.gpu

N = 100_000
THETA = 0.75


@pytest.fixture(scope="module")
def problem():
    m, x, y, z = plummer_numpy(N, "float32")  # the generator and seed of `bench.py --workload plummer100k_f32`
    ot = oracle.Tree(x, y, z, m)
    ref, stats = ot.acc_pot(0, THETA, nthreads=8, want_stats=True)
    t = rakau_amd.Octree(x, y, z, m)
    # Same tree: nodes and critical nodes of the front door's builder are the oracle's.
    assert (t.n_nodes, len(t.crit_nodes())) == (ot.n_nodes, len(ot.crit_nodes()))
    return t, ref, stats


def test_100k_cpu_engine(problem):
    t, ref, _ = problem
    scalar = t.cpu_acc_pot_u(0, THETA, flavour="scalar")
    for g, r in zip(scalar, ref):
        assert np.array_equal(g, r)  # the scalar flavour is the oracle's arithmetic in the oracle's order
    simd = t.cpu_acc_pot_u(0, THETA, flavour="auto")
    e = rel_err_vec(simd, ref)
    assert e.max() <= 1e-5 and np.median(e) <= 1e-6, (e.max(), np.median(e))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_100k_hip_path_on_every_critical_node(problem, variant):
    t, ref, stats = problem
    st = t.state()
    assert st.n_crit > 2500  # about 2.9k critical nodes: the whole tree goes through the kernels
    st.set_variant(variant)
    try:
        got = st.acc_pot(0, rakau_amd.mac_value_of(THETA, "bh", np.float32))
    finally:
        st.set_variant(0)
    assert all(np.all(np.isfinite(g)) for g in got)
    e = rel_err_vec(got, ref)
    # Identical interaction lists (critical-node grouping): rounding only. The reference's own bound is 2e-3
    # (test/ordering_acc.cpp:96).
    assert e.max() <= 2e-5 and np.median(e) <= 1e-6, (variant, e.max(), np.median(e))
    # The interaction census of the traversal equals the oracle's exactly.
    if variant == 0:
        c = st.count_interactions(rakau_amd.mac_value_of(THETA, "bh", np.float32))
        assert (c["mac"], c["com"], c["pp"], c["self"]) == (stats["w_visits"], stats["w_com"], stats["w_pp"], stats["w_self"])
