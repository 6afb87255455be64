"""Rehearsal of bench.py's multi-rank path on a 1-GPU box: two ranks launched by torch.distributed.run share
GPU 0 (RK_BENCH_SINGLE_DEVICE=1) and replicate the tree through host memory (RK_BENCH_BACKEND=gloo). Everything
but the RCCL transport is the code the driver runs at N = 2, 4, 8: export -> broadcast -> import, Morton shards,
max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_one_gpu(scaling):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RK_BENCH_SINGLE_DEVICE="1", RK_BENCH_BACKEND="gloo", RK_BENCH_SCALING=scaling)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "plummer100k_f32"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["scaling"] == scaling and d["unit"] == "Mparticles/s"
    assert "roofline" in d and "cpu_baseline" not in d
    assert "equal interaction counts" in d["config"]["sharding"]
    if scaling == "strong":
        # Both shards together evaluate every interaction of the full 100k problem.
        assert d["config"]["nparts"] == 100000 and d["config"]["nparts_per_gpu"] == 50000
        assert abs(d["interactions_per_particle"] - 1218.45) < 1.0
    else:
        # One sphere of 2 x 100k particles, 100k targets per rank.
        assert d["config"]["nparts"] == 200000 and d["config"]["nparts_per_gpu"] == 100000
        assert 1218.45 < d["interactions_per_particle"] < 1500
