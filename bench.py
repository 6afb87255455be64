#!/usr/bin/env python3
"""Benchmark of the hot path: accs_u() on a Plummer sphere (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one accs_u() call through the accelerator seam's own signature -- rk_acc_pot(): one traversal of the resident
tree for all target particles of the rank, tree and particles already in HBM, the results delivered into the CALLER'S HOST
arrays (t = wall time of one accs_u() with the tree resident and the outputs landing in host memory). The
output arrays are allocated in pinned memory through the `Allocator` parameter the reference's own overloads have
(tree.hpp:3406-3412; rakau_amd::pinned_allocator / rk_host_alloc). The same K steps with the results left in HBM
(a device-resident caller, rk_acc_pot_device) and into pageable arrays are timed next to it and reported as
`value_device_resident` / `value_host_outputs_pageable`. For N > 1 there is one rank per GPU:
either launched by torch.distributed.run (the driver's form) or -- when WORLD_SIZE is not set -- spawned by this
script itself as N child processes before anything touches a GPU (the reference needs no launcher either,
tree.hpp:3150-3240). Rank 0 builds the tree and uploads it, the device buffers are replicated with RCCL
broadcasts, and every rank traverses its contiguous Morton shard of the targets (cut at critical-node boundaries,
equal interaction counts): no data-path collective.

Default for N > 1 is the BASELINE.json metric: the SAME 4M-particle problem split over the N GPUs ("scaling":
"strong"). `--scaling weak` makes the N-GPU problem ONE sphere of N x 4M particles in one replicated tree (4M targets
per GPU); `--workload plummer64m_f32 --gpus 8` is BASELINE config 5 (64M particles sharded over 8 GPUs). The
`metric` / `config.workload` strings always state the real particle count of the run.

Rank 0 prints ONE JSON line. Besides the contract's fields it carries `roofline` (compute bound on the FP32/FP64
vector ALU: the path is rsqrt/FMA bound; reported in the contract's "mfma" class, whose peak is
the same number; the compulsory-HBM figures ride along) and, at
N = 1, `cpu_baseline` (the C++ header's own multi-threaded AVX2 CPU engine on the whole workload, plus the scalar CPU
oracle on a bounded sample, which doubles as the parity checker; both on the host cores of the same box).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Workloads: BASELINE.json configs. "workload" strings name them in the JSON line.
WORKLOADS = {
    "plummer4m_f32": dict(n=4_000_000, dtype="float32", theta=0.75, q=0, eps=0.0,
                          desc="3D fp32, 4M-particle Plummer, theta=0.75, accs_u() (README headline config)"),
    "plummer100k_f32": dict(n=100_000, dtype="float32", theta=0.75, q=0, eps=0.0,
                            desc="3D fp32, 100k-particle Plummer, theta=0.75, accs_u()"),
    "plummer4m_f32_accpot": dict(n=4_000_000, dtype="float32", theta=0.75, q=2, eps=None,
                                 desc="3D fp32, 4M-particle Plummer, theta=0.75, accs_pots_u() with softening"),
    "plummer16m_f64": dict(n=16_000_000, dtype="float64", theta=0.5, q=0, eps=0.0,
                           desc="3D fp64, 16M-particle Plummer, theta=0.5, accs_u()"),
    "plummer64m_f32": dict(n=64_000_000, dtype="float32", theta=0.75, q=0, eps=0.0,
                           desc="3D fp32, 64M-particle Plummer, theta=0.75, accs_u()"),
}

PEAK_TFLOPS = {"float32": 157.3, "float64": 78.6}  # vector ALU peaks, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def plummer_numpy(n, dtype, seed=20261002, a=1.0):
    """Plummer sphere with the transform of benchmark/common.hpp:95-124 of the reference (masses U[0.1,1.9),
    r = a / sqrt(u^(-2/3) - 1), uniform direction), drawn from numpy's PCG64 with a recorded seed."""
    rng = np.random.default_rng(seed)
    f = np.dtype(dtype).type
    m = rng.uniform(0.1, 1.9, n).astype(dtype)
    u = rng.random(n)
    u = np.clip(u, 1e-12, 1.0 - 1e-12)
    r = a / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
    lon = 2.0 * np.pi * rng.random(n)
    colat = np.arccos(np.clip(2.0 * rng.random(n) - 1.0, -1.0, 1.0))
    x = (r * np.cos(lon) * np.sin(colat)).astype(dtype)
    y = (r * np.sin(lon) * np.sin(colat)).astype(dtype)
    z = (r * np.cos(colat)).astype(dtype)
    del f
    return m, x, y, z


def usable_cpus():
    """Host cores this process may actually use: min(os.cpu_count(), cgroup v2 cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def shard_cuts(crit_ranges, nparts, world, work=None):
    """Contiguous Morton shards cut at critical-node boundaries: equal particle counts, or -- given the per-group
    interaction counts of rk_group_work() -- equal traversal work (a Plummer core particle costs ~3x a halo one)."""
    begins = crit_ranges[:, 0]
    cuts = [0]
    if work is None:
        for r in range(1, world):
            target = nparts * r // world
            i = int(np.searchsorted(begins, target, side="left"))
            cuts.append(int(begins[i]) if i < len(begins) else nparts)
    else:
        cum = np.cumsum(work.astype(np.float64))
        for r in range(1, world):
            # First group whose cumulative work (exclusive) reaches r / world of the total.
            i = int(np.searchsorted(cum, cum[-1] * r / world, side="left")) + 1 if len(cum) else 0
            cuts.append(int(begins[i]) if i < len(begins) else nparts)
    cuts.append(nparts)
    return [max(c, cuts[i - 1]) if i else c for i, c in enumerate(cuts)]


def self_launch(world):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of THIS process (which never
    initialises a GPU), with the environment torch.distributed.run would give them, and return their exit status.
    Rank 0 prints the JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # The GPU pool's host driver only supports dmabuf IPC (its own documentation: without this setting RCCL and the
        # sharing of device memory across processes fail with `hipIpcGetMemHandle: invalid argument`); the pool exports it
        # already, a copy of the script run elsewhere gets it here. Never overrides a value the caller chose.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    for q in pending:  # a dead rank leaves the others waiting in a collective: stop them
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="plummer4m_f32", choices=sorted(WORKLOADS))
    ap.add_argument("--nparts", type=int, default=None, help="override the particle count of the workload")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = default)")
    ap.add_argument("--builder", default="host", choices=["device", "host"],
                    help="where the tree of the timed state is built (host = the C++ header's builder, whose trees are "
                         "bit-identical to the CPU oracle's; device = rk_state_build, centres of mass differ by rounding)")
    ap.add_argument("--scaling", default=os.environ.get("RK_BENCH_SCALING", "strong"), choices=["strong", "weak"],
                    help="N > 1: strong = the workload's particle count in total (BASELINE metric), weak = per GPU")
    ap.add_argument("--reuse-prepass", action="store_true",
                    help="let repeated calls on the resident tree reuse the output of the supergroup pre-pass (the "
                         "library's default, +1.5 %% at 4M); by default every timed step runs the whole traversal")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pageable-leg", action="store_true",
                    help="skip the extra calls into pageable arrays after the timed steps (profiling runs: their sub-range "
                         "launches would mix into the per-kernel averages of the timed steps)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all host cores)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # No launcher: become one. Nothing in this process has touched (or will touch) a GPU.
        raise SystemExit(self_launch(args.gpus))

    # Every timed step does the whole traversal: nothing computed by an earlier step is reused (the library would
    # otherwise keep the supergroup pre-pass lists of a resident tree across calls).
    if not args.reuse_prepass:
        os.environ["RK_SUPER_CACHE"] = "0"
    import torch
    import rakau_amd
    from rakau_amd import _capi

    wl = dict(WORKLOADS[args.workload])
    if args.nparts:
        wl["n"] = args.nparts
    n, dtype, theta, q = wl["n"], wl["dtype"], wl["theta"], wl["q"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    scaling = args.scaling
    if scaling == "weak":
        n = n * world  # one sphere of world x n particles; every GPU owns n of them
    # Softening of the reference's leapfrog benchmark, eps = 0.45 * N^-0.73 (benchmark_leapfrog.cpp:218-223).
    eps = wl["eps"] if wl["eps"] is not None else 0.45 * n ** -0.73
    nres = rakau_amd.NRES[q]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # RK_BENCH_SINGLE_DEVICE=1 + RK_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path on a 1-GPU box
    # (all ranks share GPU 0, buffers are replicated through host memory). The real path is nccl (= RCCL).
    backend = os.environ.get("RK_BENCH_BACKEND", "nccl")
    single_dev = os.environ.get("RK_BENCH_SINGLE_DEVICE", "0") == "1"
    dev = local_rank if (world > 1 and not single_dev) else 0
    if dev >= torch.cuda.device_count():
        raise SystemExit("bench.py --gpus %d needs %d GPUs, %d visible (RK_BENCH_SINGLE_DEVICE=1 RK_BENCH_BACKEND=gloo "
                         "rehearses the multi-rank path on one GPU)" % (args.gpus, args.gpus, torch.cuda.device_count()))
    # Cold start, measured before anything else in this process touches the GPU: rk_init() = HIP runtime + device context +
    # the code objects of all kernel families (seconds on a box whose libraries are not yet in the page cache).
    t0 = time.perf_counter()
    _capi.check(_capi.lib().rk_init(dev))
    t_init = time.perf_counter() - t0
    torch.cuda.set_device(dev)
    dist = None
    # RK_BENCH_FORCE_COMM=1 (test knob, world == 1 only): take the multi-rank replicate branch with a communicator of one rank --
    # torch's own nccl process group, Comm.unique_id -> broadcast_object_list -> rk_comm_init -> rk_state_broadcast ->
    # close, in the order an 8-GPU run uses them -- on a 1-GPU box (RCCL refuses two ranks on one GPU).
    force_comm = world == 1 and os.environ.get("RK_BENCH_FORCE_COMM", "0") == "1"
    if world > 1 or force_comm:
        import torch.distributed as dist
        if force_comm and "MASTER_ADDR" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
            sk.close()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = _capi.lib()
    mac = "bh"
    mac_value = rakau_amd.mac_value_of(theta, mac, dtype)
    eps2 = float(np.dtype(dtype).type(eps) ** 2)

    # ---- build (rank 0) and replicate -------------------------------------------------------------------
    t_build = t_upload = 0.0
    tree = None
    if rank == 0:
        m, x, y, z = plummer_numpy(n, dtype)
        if args.builder == "device":
            # Node sums in the reference's association: the device-built tree is then bit-identical to the host-built one
            # (same MAC decisions, same interaction census), so the parity gate of the timed traversal is the same.
            rakau_amd.set_build_exact(True)
            rakau_amd.Octree(x[:1024], y[:1024], z[:1024], m[:1024], mac=mac, builder="device").close()  # warm up
        t0 = time.perf_counter()
        tree = rakau_amd.Octree(x, y, z, m, mac=mac, builder=args.builder)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        state = tree.state()
        torch.cuda.synchronize()
        t_upload = time.perf_counter() - t0
        # Device-side construction of the same tree, timed for the record.
        t_dev_build = None
        t_dev_build_exact = None
        try:
            rakau_amd.State.build(x[:1024], y[:1024], z[:1024], m[:1024], mac=mac).close()
            for exact in (False, True):
                rakau_amd.set_build_exact(exact)
                t0 = time.perf_counter()
                sb = rakau_amd.State.build(x, y, z, m, mac=mac)
                dt = time.perf_counter() - t0
                sb.close()
                if exact:
                    t_dev_build_exact = dt
                else:
                    t_dev_build = dt
        except Exception as e:  # pragma: no cover
            t_dev_build = str(e)
        finally:
            rakau_amd.set_build_exact(args.builder == "device")
    replicate_via = None
    lib_comm = None
    lib_comm_used = False
    if (world > 1 or force_comm) and backend == "nccl":
        # The replicate step lives in the library: rk_comm_* + rk_state_broadcast (ncclBroadcast of the meta block and of
        # every buffer of the state, RCCL over xGMI). torch.distributed only ships the 128-byte communicator id -- and is
        # the fallback (same buffers through dist.broadcast) if RCCL cannot be bound by the library on SOME rank. That is
        # agreed on BEFORE the collective ncclCommInitRank: every rank probes with rk_comm_unique_id (dlopen + a local RCCL
        # call, not collective) and the outcomes are all-reduced, so all ranks take the same branch. A failure inside the
        # collective init itself is not recoverable (the other ranks are inside it) and ends the run.
        from rakau_amd.state import Comm
        ok, my_id = 1, None
        try:
            my_id = Comm.unique_id()
        except Exception as e:  # pragma: no cover
            ok = 0
            print("rank %d: RCCL not usable from the library (rk_comm_unique_id: %s)" % (rank, e), file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            payload = [my_id if rank == 0 else None]
            dist.broadcast_object_list(payload, src=0)
            lib_comm = Comm(world, payload[0], rank, dev)
    if lib_comm is not None:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        state = rakau_amd.State.broadcast(state if rank == 0 else None, 0, rank, dev, lib_comm)
        torch.cuda.synchronize()
        t_replicate = time.perf_counter() - t0
        lib_comm.close()
        lib_comm_used = True
        replicate_via = "rk_state_broadcast (RCCL, inside the library)"
    elif world > 1 or force_comm:
        # rk_state_export -> torch.distributed broadcast -> rk_state_import: the one-GPU rehearsal of the multi-rank path
        # (gloo: the buffers travel through host memory) and the fallback of the branch above (nccl: device to device).
        payload = [None]
        if rank == 0:
            ptrs, nbytes, meta = state.export()
            payload = [(nbytes, meta)]
        dist.broadcast_object_list(payload, src=0)
        nbytes, meta = payload[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bufs = [torch.empty(max(b, 1), dtype=torch.uint8, device="cuda") for b in nbytes]
        if rank == 0:
            for t, p, b in zip(bufs, ptrs, nbytes):
                _capi.check(lib.rk_device_memcpy(t.data_ptr(), p, b, dev))
        for t in bufs:
            if backend == "nccl":
                dist.broadcast(t, src=0)
            else:
                h = t.cpu()
                dist.broadcast(h, src=0)
                t.copy_(h)
        torch.cuda.synchronize()
        t_replicate = time.perf_counter() - t0
        if rank != 0:
            state = rakau_amd.State.from_buffers(dev, [t.data_ptr() for t in bufs], nbytes, meta)
        del bufs
        replicate_via = "rk_state_export / torch.distributed broadcast (%s) / rk_state_import" % backend
    else:
        t_replicate = 0.0
    if args.variant:
        state.set_variant(args.variant)

    crit = state.crit_ranges()
    # Shards of equal WORK (integer census on the replicated tree: every rank derives the same cuts, no exchange).
    balance = os.environ.get("RK_BENCH_BALANCE", "work")
    work = state.group_work(mac_value) if (world > 1 and balance == "work") else None
    cuts = shard_cuts(crit, state.nparts, world, work)
    p_begin, p_end = cuts[rank], cuts[rank + 1]
    n_local = p_end - p_begin

    tdt = torch.float32 if dtype == "float32" else torch.float64
    outs = [torch.zeros(max(n_local, 1), dtype=tdt, device="cuda") for _ in range(nres)]
    d_ptrs = [o.data_ptr() for o in outs]
    stream = torch.cuda.current_stream().cuda_stream
    # Host output arrays of the timed call, allocated the way the reference's `Allocator` overloads allow (pinned).
    pin_out = [rakau_amd.pinned_empty(max(n_local, 1), dtype) for _ in range(nres)]

    def step():
        """One accs_u() through the seam: rk_acc_pot() for this rank's Morton range, results into host arrays. Blocking."""
        state.acc_pot(q, mac_value, eps2=eps2, p_begin=p_begin, p_end=p_end, out=pin_out, offset_output=False)

    def step_dev():
        """The same traversal for a device-resident caller: results left in HBM, enqueued on the stream."""
        state.acc_pot_device(q, mac_value, d_ptrs, G=1.0, eps2=eps2, p_begin=p_begin, p_end=p_end,
                             offset_output=False, stream=stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Order of the measurements: the interaction census of this rank's range (what the roofline divides by), then the
    # kernel-only time of a step (HIP events recorded by the library on the launch stream, one synchronised call at a
    # time), then the W warmup steps and the K timed steps. The first two also take the GPU out of the idle power state the
    # host-side tree construction left it in: after idling a 4M step needs 50-80 ms of load before its duration settles
    # (2.80, 2.71, 2.65, 2.60, 2.53, 2.49, 2.45, 2.43, 2.43, 2.38, ... 2.33 ms; RK_BENCH_DEBUG=1 prints the series), more
    # than W = 3-5 steps provide. Hence at least 5 calls and 80 ms of kernel time here; the median of the last (up to)
    # 10 is `kernel_ms`. Workloads whose calls take tens of milliseconds need, and get, no more than that.
    # The very first traversal call on this state (launch-path set-up, scratch allocation, the supergroup pre-pass).
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_dev()
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t0) * 1e3
    census = state.count_interactions(mac_value, p_begin, p_end)
    inter_local = census["com"] + census["pp"] + census["self"]

    def settle(fn):
        kms, busy_ms = [], 0.0
        while len(kms) < 5 or (busy_ms < 80.0 and len(kms) < 1000):
            fn()
            kms.append(state.last_kernel_ms())
            busy_ms += kms[-1] if len(kms) > 2 else 0.0  # the first calls carry one-off initialisation
        if os.environ.get("RK_BENCH_DEBUG"):
            print("kernel ms of a pre-loop:", " ".join("%.3f" % v for v in kms), file=sys.stderr)
        return float(np.median(kms[-min(10, len(kms) - 2):]))

    kernel_ms_dev = settle(step_dev)
    # One burst of K calls without synchronisation in between: the first time K launches are in flight the HIP runtime
    # grows its signal / command pools (+0.05 ms per step on the first burst of 20, none afterwards; irrelevant, and
    # skipped, when a call takes longer than 5 ms).
    if kernel_ms_dev < 5.0:
        for _ in range(args.steps):
            step_dev()
    barrier()
    # (1) The device-resident caller: K calls back to back, results left in HBM, without the library's two timing events
    # per call (barrier packets between consecutive calls). Reported as `value_device_resident`.
    state.set_timing(False)
    for _ in range(args.warmup):
        step_dev()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_dev()
    barrier()
    elapsed_dev = time.perf_counter() - t0
    state.set_timing(True)
    # (2) The seam's own call, results into (pinned) host arrays: kernel time of that form (the epilogue stores cross
    # PCIe), then THE timed region of the contract: W warmup steps, K timed steps, barrier + synchronize on both sides.
    kernel_ms = settle(step)
    barrier()
    state.set_timing(False)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    state.set_timing(True)
    if os.environ.get("RK_BENCH_DEBUG"):
        for _ in range(4):
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            print("another %d steps: %.4f ms per step" % (args.steps, (time.perf_counter() - t1) / args.steps * 1e3), file=sys.stderr)

    if dist is not None:
        rdev = "cuda" if backend == "nccl" else "cpu"
        red = torch.tensor([elapsed, kernel_ms, elapsed_dev, kernel_ms_dev], dtype=torch.float64, device=rdev)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max, elapsed_dev, kernel_ms_dev_max = (float(v) for v in red)
        tot = torch.tensor([float(inter_local), float(census["mac"])], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        inter_total, mac_total = float(tot[0]), float(tot[1])
    else:
        kernel_ms_max, kernel_ms_dev_max = kernel_ms, kernel_ms_dev
        inter_total, mac_total = float(inter_local), float(census["mac"])

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = n / (elapsed / args.steps) / 1e6
    fsz = np.dtype(dtype).itemsize
    flop_per_inter = 19 if q == 0 else (21 if q == 2 else 12)
    # Roofline of the dominant (traversal) launch on this rank: algorithmic flops / measured kernel time.
    flops_launch = inter_local * flop_per_inter
    achieved_tflops = flops_launch / (kernel_ms * 1e-3) / 1e12
    n_nodes = state.tree_size
    bytes_launch = n_local * (4 * fsz + nres * fsz) + n_nodes * (5 * fsz + 12)
    hbm_gbs = bytes_launch / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per step from PMC counters (tools/measure_traffic.sh; separate rocprofv3 --pmc passes of this very
    # command, gfx950 correction applied), if a measurement for this workload has been committed.
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get("workload") == args.workload and world == 1 and not args.nparts:
            traffic = int(tj["hbm_bytes_per_step"])
    except Exception:
        pass
    def count_str(k):
        return "%dM" % (k // 1_000_000) if k % 1_000_000 == 0 else ("%dk" % (k // 1000) if k % 1000 == 0 else str(k))

    call = {0: "accs_u()", 1: "pots_u()", 2: "accs_pots_u()"}[q]
    fp_name = "fp32" if dtype == "float32" else "fp64"
    # Always states the REAL particle count of the run (n = whole problem, all GPUs together).
    what = "%s %s Plummer %s theta=%g%s" % (call, count_str(n), fp_name, theta, " with softening" if eps else "")
    workload = "3D %s, %s-particle Plummer, theta=%g, %s%s" % (fp_name, count_str(n), theta, call,
                                                             " with softening" if eps else "")
    # How the replicas were really made in THIS run (the library's RCCL broadcast, or the export / torch.distributed / import
    # path with the transport that carried it).
    if lib_comm_used:
        transport = "tree replicated by RCCL broadcast (rk_state_broadcast)"
    else:
        transport = "tree replicated through torch.distributed broadcast (%s%s)" % (
            backend, " = RCCL" if backend == "nccl" else ": host memory, rehearsal on shared GPU(s)")
    if world > 1:
        workload += (", Morton-sharded across %d GPUs (%s targets per GPU), %s"
                     % (world, count_str(n // world) if n % world == 0 else "~%d" % (n // world), transport))
    line = {
        "metric": "Mparticles/s, " + what,
        "value": round(value, 2),
        "unit": "Mparticles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32" if dtype == "float32" else "f64",
        "data": "synthetic Plummer sphere (numpy PCG64 seed 20261002), tree built on %s, resident in HBM; `value` = the "
                "seam's own call, rk_acc_pot() into the caller's host arrays (SURVEY 8(d) t), arrays allocated with the "
                "reference's Allocator overload in pinned memory (rakau_amd::pinned_allocator / rk_host_alloc); "
                "`value_device_resident` = results left in HBM (rk_acc_pot_device), `value_host_outputs_pageable` = "
                "plain pageable arrays" % args.builder,
        "config": {"workload": workload, "workload_key": args.workload,
                   "nparts": n, "nparts_per_gpu": n // world, "theta": theta, "q": q, "eps": eps, "max_leaf_n": 16,
                   "ncrit": 128, "mac": mac, "nodes": n_nodes, "critical_nodes": int(state.n_crit),
                   "sharding": ("contiguous Morton range per GPU (equal %s), %s" % ("interaction counts" if work is not None else "particle counts", transport)) if world > 1
                   else "single GPU", "kernel_variant": args.variant,
                   "prepass_reused_across_steps": bool(args.reuse_prepass)},
        "kernel_ms": round(kernel_ms_max, 4),
        # The same K steps for a device-resident caller (results left in HBM; rk_acc_pot_device, calls enqueued back to back).
        "value_device_resident": round(n / (elapsed_dev / args.steps) / 1e6, 2),
        "ms_per_step_device_resident": round(elapsed_dev / args.steps * 1e3, 4),
        "kernel_ms_device_resident": round(kernel_ms_dev_max, 4),
        "interactions_per_particle": round(inter_total / n, 2),
        "mac_evals_per_particle": round(mac_total / n, 2),
        "roofline": {
            # Compute-bound class of the contract ("hbm" | "mfma"). The kernel issues no MFMA instructions: the work is
            # distance + rsqrt per pair on the vector ALU, whose peak is the same 157.3 / 78.6 TFLOP/s
            # as the dense f32 / f64 matrix-core peak.
            "bound": "mfma", "bound_detail": "compute bound on the vector ALU (rsqrt/FMA per pair; no MFMA instructions)",
            "achieved": round(achieved_tflops, 3), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
            "frac": round(achieved_tflops / PEAK_TFLOPS[dtype], 4), "traffic": traffic,
            # `traffic` is a committed PMC figure of this very command (profiles/traffic.json, separate rocprofv3 --pmc
            # passes), not something this run measured; null when no figure for this workload has been committed.
            "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc measurement of this command, averaged over its launches -- "
                              "results left in HBM and results written through PCIe alike; not measured in this run)"
                              if traffic is not None else None,
            "flop_per_interaction": flop_per_inter, "interactions_per_launch": int(inter_local),
            "kernel_ms": round(kernel_ms, 4),
            "hbm": {"achieved": round(hbm_gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(hbm_gbs / PEAK_HBM_GBS, 5), "algorithmic_bytes_per_launch": int(bytes_launch)},
        },
        "host": {"builder": args.builder, "tree_build_s": round(t_build, 3), "upload_s": round(t_upload, 3),
                 # Cold start: rk_init (first GPU touch of the process), the first state of the process, its first call.
                 "rk_init_s": round(t_init, 3), "state_create_cold_s": round(t_upload, 3),
                 "first_call_ms": round(first_call_ms, 3),
                 "replicate_s": round(t_replicate, 4), "replicate_via": replicate_via,
                 "device_build_s": round(t_dev_build, 4) if isinstance(t_dev_build, float) else t_dev_build,
                 "device_build_exact_s": round(t_dev_build_exact, 4) if isinstance(t_dev_build_exact, float) else None},
        "reference_published": {"best_cpu_2xXeon6148_Mps": 48.8, "V100_Mps": 42.1, "RX570_rocm_path_Mps": 15.6,
                                "note": "README.md:42-49 of the reference, single cold calls on other hardware"},
    }

    # The same call into plain pageable arrays (what a tree.hpp user with std::vector<F> outputs pays per accs_u() on a
    # resident tree: the results go through the library's pinned staging buffer and are delivered by host threads).
    try:
        if world == 1 and not args.no_pageable_leg:
            host_out = [np.zeros(n, dtype=dtype) for _ in range(nres)]
            def settle_and_time(arrays):
                ts, busy = [], 0.0
                while len(ts) < 7 or (busy < 0.08 and len(ts) < 200):
                    t0 = time.perf_counter()
                    state.acc_pot(q, mac_value, eps2=eps2, out=arrays)
                    ts.append(time.perf_counter() - t0)
                    busy += ts[-1] if len(ts) > 2 else 0.0
                return float(np.median(ts[-min(10, len(ts) - 2):]))
            t_host = settle_and_time(host_out)
            line["host"]["kernel_ms_host_outputs_pageable"] = round(state.last_kernel_ms(), 4)
            line["value_host_outputs_pageable"] = round(n / t_host / 1e6, 2)
            line["ms_per_call_host_outputs_pageable"] = round(t_host * 1e3, 4)
            line["host"]["pinned_equals_pageable"] = bool(all(np.array_equal(a, b) for a, b in zip(host_out, pin_out)))
            line["host"]["pinned_equals_device_resident"] = bool(all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(pin_out, outs)))
    except Exception as e:  # pragma: no cover
        line["host"]["acc_pot_host_outputs_error"] = str(e)

    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(tree, m, x, y, z, mac, theta, eps, q, n, args.cpu_threads, outs, p_begin)

    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(tree, m, x, y, z, mac, theta, eps, q, n, threads, gpu_outs, p_begin):
    """CPU baseline on the host cores of the GPU box, same inputs, same tree parameters:
    * `value`: the CPU engine of the C++ header (include/rakau_amd/cpu_engine.hpp: critical-node tasks on std::threads,
      AVX-512 (where the CPU has it) or AVX2 batches of targets, fp32 rsqrt + Newton step -- a re-implementation of the reference's TBB + xsimd engine,
      which cannot be built here; it is also what the host share of kwargs::split runs), whole workload, best of 2;
    * `oracle_scalar_port`: the CPU oracle (scalar restatement of the reference's engine) on a bounded sample of
      critical nodes (~5 s), which is also the checker: parity of the timed GPU result and of the CPU engine against it."""
    import oracle
    threads = threads or usable_cpus()
    ts = []
    eng = None
    for _ in range(2):
        t0 = time.perf_counter()
        eng = tree.cpu_acc_pot_u(q, theta, eps=eps, nthreads=threads)
        ts.append(time.perf_counter() - t0)
        if ts[-1] > 15.0:
            break
    dt = min(ts)
    from rakau_amd import _capi
    isa = "AVX-512" if _capi.lib().rk_cpu_engine_run(None) == 0 else "AVX2"
    out = {"value": round(n / dt / 1e6, 3), "unit": "Mparticles/s", "cores": threads, "kind": "port", "flavour": "simd", "isa": isa,
           "sample": "whole workload (%d particles), best of %d calls, %.2f s" % (n, len(ts), dt),
           "note": "rakau_amd's own CPU engine (std::thread + AVX-512 / AVX2 batches, the engine behind split = {cpu, ...}); a re-implementation, "
                   "not the reference's TBB + xsimd build (not buildable here); published reference figure: 48.8 Mparticles/s "
                   "on 2 x Xeon Gold 6148 (README.md:42-49)"}
    ot = oracle.Tree(x, y, z, m, mac=mac)
    crit = ot.crit_nodes()
    ncrit = len(crit)
    probe = max(threads * 16, ncrit // 256)
    t0 = time.perf_counter()
    ot.acc_pot(q, theta, eps=eps, nthreads=threads, c_begin=0, c_end=probe)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    sample = int(min(ncrit, max(probe, rate * 5.0)))
    t0 = time.perf_counter()
    ref = ot.acc_pot(q, theta, eps=eps, nthreads=threads, c_begin=0, c_end=sample)
    dt = time.perf_counter() - t0
    parts = int(crit[sample - 1, 2])
    out["oracle_scalar_port"] = {"value": round(parts / dt / 1e6, 3), "unit": "Mparticles/s", "cores": threads, "kind": "port",
                                 "sample": "critical nodes [0, %d) of %d = particles [0, %d) of %d, %.2f s"
                                           % (sample, ncrit, parts, n, dt)}
    # Parity against the oracle on the sample (vector norm, as SURVEY 8(d) Gate A).
    try:
        def vec_err(a):
            g = np.stack([np.asarray(v[:parts], dtype=np.float64) for v in a[:3]], axis=1)
            r = np.stack([np.asarray(v[:parts], dtype=np.float64) for v in ref[:3]], axis=1)
            den = np.linalg.norm(r, axis=1)
            den[den == 0] = 1.0
            return np.linalg.norm(g - r, axis=1) / den
        if q in (0, 2):
            if p_begin == 0:
                err = vec_err([o[:parts].cpu().numpy() for o in gpu_outs[:3]])
                out["parity_max_rel_err"] = float(err.max())
                out["parity_median_rel_err"] = float(np.median(err))
            err = vec_err(eng)
            out["cpu_engine_vs_oracle_max_rel_err"] = float(err.max())
    except Exception as e:  # pragma: no cover
        out["parity_error"] = str(e)
    return out


if __name__ == "__main__":
    main()
