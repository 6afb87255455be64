"""Rehearsal of bench.py's multi-rank path on a 1-GPU box: two ranks share GPU 0 (RK_BENCH_SINGLE_DEVICE=1) and
replicate the tree through host memory (RK_BENCH_BACKEND=gloo). Everything but the RCCL transport is the code the
driver runs at N = 2, 4, 8: export -> broadcast -> import, Morton shards, max-over-ranks timing, one JSON line from
rank 0. Two launch forms: `python bench.py --gpus 2` by itself (bench.py spawns its ranks, like the reference needs no
launcher: tree.hpp:3150-3240) and the driver's torch.distributed.run form."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(launcher, extra):
    env = dict(os.environ, RK_BENCH_SINGLE_DEVICE="1", RK_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RK_BENCH_SCALING"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload",
            "plummer100k_f32"] + extra
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("launcher,scaling", [("self", None), ("self", "weak"), ("torchrun", "strong")])
def test_two_ranks_one_gpu(launcher, scaling):
    d = run_bench(launcher, ["--scaling", scaling] if scaling else [])
    scaling = scaling or "strong"  # the default is the BASELINE metric: same problem, more GPUs
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["scaling"] == scaling and d["unit"] == "Mparticles/s"
    assert "roofline" in d and "cpu_baseline" not in d
    assert "equal interaction counts" in d["config"]["sharding"]
    if scaling == "strong":
        # Both shards together evaluate every interaction of the full 100k problem.
        assert d["config"]["nparts"] == 100000 and d["config"]["nparts_per_gpu"] == 50000
        assert abs(d["interactions_per_particle"] - 1218.45) < 1.0
        assert "100k Plummer" in d["metric"] and "100k-particle" in d["config"]["workload"]
    else:
        # One sphere of 2 x 100k particles, 100k targets per rank: the labels say 200k, never 100k.
        assert d["config"]["nparts"] == 200000 and d["config"]["nparts_per_gpu"] == 100000
        assert 1218.45 < d["interactions_per_particle"] < 1500
        assert "200k Plummer" in d["metric"] and "200k-particle" in d["config"]["workload"]
    assert "across 2 GPUs" in d["config"]["workload"]
    # The labels name the transport that really carried the replicas in this run (gloo through host memory), not RCCL.
    assert "torch.distributed broadcast (gloo" in d["config"]["workload"] and "RCCL" not in d["config"]["workload"]
    assert "gloo" in d["host"]["replicate_via"] and "gloo" in d["config"]["sharding"]
    # `value` is the seam's own call (results into host arrays); the device-resident rate rides along.
    assert d["value_device_resident"] > 0 and d["kernel_ms_device_resident"] > 0


def test_shard_union_through_the_product():
    """Two ranks (gloo, sharing the GPU): replicate the state as bench.py does, traverse one Morton shard each through
    the C ABI, and compare the union of the shards with the full-range result on rank 0 -- bit for bit, compact host
    outputs and ordered device outputs (tests/multirank_product_worker.py)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_product_worker.py"), "300000"],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    assert "SHARD_UNION_EQUALS_FULL True ORDERED True" in outs[0][0], outs[0][0]


def test_bench_one_rank_takes_the_library_rccl_branch():
    """RK_BENCH_FORCE_COMM=1: `bench.py --gpus 1` runs the replicate step of the 8-GPU run with a communicator of one rank, in
    the order the 8-GPU run uses -- torch's own nccl process group, the RCCL probe agreed by all-reduce, Comm.unique_id ->
    broadcast_object_list -> rk_comm_init -> rk_state_broadcast -> close -- next to each other in one process."""
    env = dict(os.environ, RK_BENCH_FORCE_COMM="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RK_BENCH_BACKEND", "RK_BENCH_SINGLE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload",
           "plummer100k_f32", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["host"]["replicate_via"].startswith("rk_state_broadcast (RCCL")
    assert d["n_gpus"] == 1 and d["value"] > 0 and abs(d["interactions_per_particle"] - 1218.45) < 1.0
    assert d["host"]["pinned_equals_pageable"] and d["host"]["pinned_equals_device_resident"]
