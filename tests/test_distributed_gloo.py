"""Multi-GPU sharding logic exercised with two CPU processes (gloo): contiguous Morton shards cut at
critical-node boundaries tile the particle range, every rank traverses only its shard against the
replicated tree, and the concatenation equals the single-process result bit for bit (no reduction step)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    import oracle
    from bench import shard_cuts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # Rank 0 owns the inputs; the (sorted) particle arrays are replicated with broadcasts, like the device
    # buffers are over RCCL in bench.py.
    if rank == 0:
        m, x, y, z = oracle.plummer(n, np.float32)
        parts = torch.from_numpy(np.stack([x, y, z, m]))
    else:
        parts = torch.empty((4, n), dtype=torch.float32)
    dist.broadcast(parts, src=0)
    x, y, z, m = (parts[i].numpy() for i in range(4))
    tree = oracle.Tree(x, y, z, m)
    crit = tree.crit_nodes()
    ranges = np.stack([crit[:, 1], crit[:, 2]], axis=1).astype(np.int64)
    cuts = shard_cuts(ranges, n, world)
    assert cuts[0] == 0 and cuts[-1] == n and all(a <= b for a, b in zip(cuts, cuts[1:]))
    begins = set(int(b) for b in ranges[:, 0]) | {n}
    assert all(c in begins for c in cuts)
    # Critical nodes of my shard.
    c0 = int(np.searchsorted(ranges[:, 0], cuts[rank]))
    c1 = int(np.searchsorted(ranges[:, 0], cuts[rank + 1]))
    res = tree.acc_pot(0, 0.75, nthreads=2, c_begin=c0, c_end=c1)
    mine = torch.from_numpy(np.stack([r[cuts[rank]:cuts[rank + 1]] for r in res]))
    # Gather the shards on rank 0 (outputs are disjoint: concatenation, not reduction).
    sizes = [cuts[i + 1] - cuts[i] for i in range(world)]
    if rank == 0:
        bufs = [torch.empty((3, s), dtype=torch.float32) for s in sizes]
        bufs[0] = mine
        for r in range(1, world):
            dist.recv(bufs[r], src=r)
        full = np.concatenate([b.numpy() for b in bufs], axis=1)
        ref = np.stack(tree.acc_pot(0, 0.75, nthreads=2))
        np.save(os.path.join(out_dir, "ok.npy"), np.array([np.array_equal(full, ref), abs(max(sizes) - min(sizes))]))
    else:
        dist.send(mine, dst=0)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    world, n = 2, 30000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    ok = np.load(tmp_path / "ok.npy")
    assert ok[0] == 1  # bit-identical to the unsharded traversal
    assert ok[1] <= 256  # balanced to within a couple of critical nodes


def test_shard_cuts_properties():
    sys.path.insert(0, ROOT)
    from bench import shard_cuts
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 129, size=5000)
    ends = np.cumsum(sizes)
    ranges = np.stack([ends - sizes, ends], axis=1)
    n = int(ends[-1])
    for world in (1, 2, 3, 4, 8, 16):
        cuts = shard_cuts(ranges, n, world)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == n
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        assert all(c == n or c in set(ranges[:, 0]) for c in cuts)
        assert max(b - a for a, b in zip(cuts, cuts[1:])) <= n // world + 128


def test_work_balanced_shard_cuts():
    """Cuts by per-group work (rk_group_work): still on critical-node boundaries, every shard within one group's work of
    the mean, degenerate inputs (one group, more ranks than groups, zero work) handled."""
    sys.path.insert(0, ROOT)
    from bench import shard_cuts
    rng = np.random.default_rng(1)
    sizes = rng.integers(1, 129, size=4000)
    ends = np.cumsum(sizes)
    ranges = np.stack([ends - sizes, ends], axis=1)
    n = int(ends[-1])
    work = (sizes * rng.integers(500, 4000, size=sizes.size)).astype(np.uint64)
    for world in (1, 2, 3, 8):
        cuts = shard_cuts(ranges, n, world, work)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == n
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        assert all(c == n or c in set(ranges[:, 0]) for c in cuts)
        g = np.searchsorted(ranges[:, 0], cuts[:-1]).tolist() + [len(sizes)]
        per = [float(work[a:b].sum()) for a, b in zip(g, g[1:])]
        assert max(per) - min(per) <= 2 * float(work.max())
    one = np.array([[0, 50]])
    assert shard_cuts(one, 50, 4, np.array([7], dtype=np.uint64)) == [0, 50, 50, 50, 50]
    assert shard_cuts(ranges[:3], int(ends[2]), 2, np.zeros(3, dtype=np.uint64))[-1] == int(ends[2])
