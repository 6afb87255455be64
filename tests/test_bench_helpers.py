"""Host logic of bench.py that needs no GPU: the Morton shards of the multi-GPU path (contiguous, cut at critical-node
boundaries, equal particle counts or equal work) and the synthetic Plummer generator."""
import numpy as np

import bench


def _crit(sizes):
    ends = np.cumsum(sizes)
    begins = ends - sizes
    return np.stack([begins, ends], axis=1).astype(np.int64)


def test_shard_cuts_cover_the_range_at_critical_node_boundaries():
    rng = np.random.default_rng(5)
    sizes = rng.integers(1, 129, 5000)
    crit = _crit(sizes)
    n = int(sizes.sum())
    for world in (1, 2, 3, 4, 8):
        cuts = bench.shard_cuts(crit, n, world)
        assert cuts[0] == 0 and cuts[-1] == n and len(cuts) == world + 1
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        assert set(cuts[1:-1]) <= set(crit[:, 0].tolist())
        # equal particle counts to within one critical node
        assert max(b - a for a, b in zip(cuts, cuts[1:])) - min(b - a for a, b in zip(cuts, cuts[1:])) <= 2 * 128


def test_shard_cuts_of_equal_work():
    rng = np.random.default_rng(6)
    sizes = rng.integers(1, 129, 4000)
    crit = _crit(sizes)
    n = int(sizes.sum())
    # work per node: a core (first nodes) three times as expensive per particle as the halo
    work = (sizes * np.where(np.arange(len(sizes)) < 1000, 3000, 1000)).astype(np.uint64)
    for world in (2, 4, 8):
        cuts = bench.shard_cuts(crit, n, world, work)
        assert cuts[0] == 0 and cuts[-1] == n and set(cuts[1:-1]) <= set(crit[:, 0].tolist())
        idx = [int(np.searchsorted(crit[:, 0], c)) for c in cuts[:-1]] + [len(sizes)]
        per = [float(work[a:b].sum()) for a, b in zip(idx, idx[1:])]
        assert max(per) / (sum(per) / world) < 1.02
        # ... which particle counts alone do not give
        eq = bench.shard_cuts(crit, n, world)
        idx = [int(np.searchsorted(crit[:, 0], c)) for c in eq[:-1]] + [len(sizes)]
        per_eq = [float(work[a:b].sum()) for a, b in zip(idx, idx[1:])]
        assert max(per_eq) / (sum(per_eq) / world) > 1.1
    # more ranks than critical nodes: trailing shards are empty, nothing is lost
    tiny = _crit(np.array([5, 7, 9]))
    cuts = bench.shard_cuts(tiny, 21, 8)
    assert cuts[0] == 0 and cuts[-1] == 21 and all(a <= b for a, b in zip(cuts, cuts[1:]))


def test_plummer_generator_is_reproducible_and_plummer_like():
    m, x, y, z = bench.plummer_numpy(20000, "float32")
    m2, x2, _, _ = bench.plummer_numpy(20000, "float32")
    assert np.array_equal(m, m2) and np.array_equal(x, x2)
    assert m.dtype == np.float32 and 0.1 <= m.min() and m.max() < 1.9 and abs(m.mean() - 1.0) < 0.02
    r = np.sqrt(x.astype(np.float64) ** 2 + y.astype(np.float64) ** 2 + z.astype(np.float64) ** 2)
    # half-mass radius of a Plummer sphere with a = 1: 1 / sqrt(2^(2/3) - 1) = 1.305
    assert abs(np.median(r) - 1.305) < 0.05
