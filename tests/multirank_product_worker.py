"""Worker of tests/test_gpu_bench_multirank.py::test_shard_union_through_the_product: one of WORLD_SIZE ranks sharing GPU 0
(gloo transport). Rank 0 builds the tree and the state, the device buffers are replicated exactly as bench.py does
(rk_state_export -> broadcast -> rk_state_import), every rank traverses its Morton shard of equal work through the C ABI,
rank 0 gathers the shards and compares their union with its own full-range result, bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rakau_amd  # noqa: E402
from rakau_amd import _capi  # noqa: E402
from bench import plummer_numpy, shard_cuts  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n = int(sys.argv[1])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    lib = _capi.lib()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    if rank == 0:
        m, x, y, z = plummer_numpy(n, "float32")
        tree = rakau_amd.Octree(x, y, z, m)
        state = tree.state()
        state.set_perm(tree.perm())
        ptrs, nbytes, meta = state.export()
        payload = [(nbytes, meta)]
    else:
        payload = [None]
    dist.broadcast_object_list(payload, src=0)
    nbytes, meta = payload[0]
    bufs = [torch.empty(max(b, 1), dtype=torch.uint8, device="cuda") for b in nbytes]
    if rank == 0:
        for t, p, b in zip(bufs, ptrs, nbytes):
            _capi.check(lib.rk_device_memcpy(t.data_ptr(), p, b, 0))
    for t in bufs:
        h = t.cpu()
        dist.broadcast(h, src=0)
        t.copy_(h)
    torch.cuda.synchronize()
    if rank != 0:
        state = rakau_amd.State.from_buffers(0, [t.data_ptr() for t in bufs], nbytes, meta)
    cuts = shard_cuts(state.crit_ranges(), state.nparts, world, state.group_work(mv))
    b, e = cuts[rank], cuts[rank + 1]
    mine = np.stack(state.acc_pot(2, mv, eps2=1e-6, p_begin=b, p_end=e, offset_output=False))
    # Ordered (original-order) device outputs work on the replica too: the permutation travelled with the buffers.
    outs = [torch.zeros(state.nparts, dtype=torch.float32, device="cuda") for _ in range(4)]
    state.acc_pot_device(2, mv, [o.data_ptr() for o in outs], eps2=1e-6, p_begin=b, p_end=e, ordered=True)
    torch.cuda.synchronize()
    ordered = np.stack([o.cpu().numpy() for o in outs])
    if rank == 0:
        full = np.stack(state.acc_pot(2, mv, eps2=1e-6))
        perm = tree.perm().astype(np.int64)
        union = np.empty_like(full)
        union[:, b:e] = mine
        ord_sum = ordered.copy()
        for r in range(1, world):
            shard = torch.empty((4, cuts[r + 1] - cuts[r]), dtype=torch.float32)
            dist.recv(shard, src=r)
            union[:, cuts[r]:cuts[r + 1]] = shard.numpy()
            o = torch.empty((4, state.nparts), dtype=torch.float32)
            dist.recv(o, src=r)
            ord_sum += o.numpy()  # disjoint supports: every element is written by exactly one rank
        ok = bool(np.array_equal(union, full))
        exp = np.empty_like(full)
        exp[:, perm] = full
        ok_o = bool(np.array_equal(ord_sum, exp))
        print("SHARD_UNION_EQUALS_FULL %s ORDERED %s cuts %s" % (ok, ok_o, cuts))
    else:
        dist.send(torch.from_numpy(mine), dst=0)
        dist.send(torch.from_numpy(ordered), dst=0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
