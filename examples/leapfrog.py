#!/usr/bin/env python3
"""GPU-resident kick-drift-kick leapfrog on a Plummer sphere: the workload of the reference's
benchmark/benchmark_leapfrog.cpp (Plummer model with velocities clipped at 10 core radii, Athanassoula softening
0.45 * N**-0.73, G = M = 1; per step: half kick, drift, tree rebuild, accelerations, half kick) with every array
living in HBM for the whole run:

    v += a * dt/2 ; x += v * dt           torch elementwise kernels
    rk_state_rebuild_device(x, y, z, m)    Morton sort + tree build on the GPU, buffers recycled
    rk_acc_pot_device(..., RK_OUT_ORDERED) traversal, results scattered back to the caller's particle order
    v += a * dt/2

Positions and velocities stay in the caller's ORIGINAL order: the ordered output mode replaces the reference's
last_perm() bookkeeping of the velocity arrays (benchmark_leapfrog.cpp:263-281, 372-381).

    python examples/leapfrog.py --nparts 1000000 --steps 20 [--track-integrals] [--fp_type double]

Prints one JSON line: steps/s and the per-step split (rebuild / traversal / integrator).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def plummer_with_velocities(n, a=1.0, seed=0, dtype=np.float32):
    """Plummer sphere, G = M = 1, equal masses, isotropic velocities from the distribution function by rejection
    sampling of q = v / v_esc with density q^2 (1 - q^2)^(7/2) (same model as benchmark_leapfrog.cpp:50-104), clipped
    at 10 core radii (:190-214). Returns x, y, z, vx, vy, vz (float64 -> dtype)."""
    rng = np.random.default_rng(seed)
    r = a / np.sqrt(rng.random(n) ** (-2.0 / 3.0) - 1.0)
    ct = rng.uniform(-1.0, 1.0, n)
    st = np.sqrt(1.0 - ct * ct)
    ph = rng.uniform(0.0, 2.0 * math.pi, n)
    x, y, z = r * st * np.cos(ph), r * st * np.sin(ph), r * ct
    q = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        xs = rng.random(todo.size)
        ys = rng.uniform(0.0, 0.1, todo.size)
        ok = ys <= xs * xs * (1.0 - xs * xs) ** 3.5
        q[todo[ok]] = xs[ok]
        todo = todo[~ok]
    v = q * math.sqrt(2.0 / a) * (1.0 + r * r / (a * a)) ** -0.25
    ct = rng.uniform(-1.0, 1.0, n)
    st = np.sqrt(1.0 - ct * ct)
    ph = rng.uniform(0.0, 2.0 * math.pi, n)
    vx, vy, vz = v * st * np.cos(ph), v * st * np.sin(ph), v * ct
    keep = x * x + y * y + z * z < 100.0 * a * a
    return tuple(np.ascontiguousarray(c[keep], dtype=dtype) for c in (x, y, z, vx, vy, vz))


class Leapfrog:
    """State of one run; step() advances by dt. All tensors are on `device`, in the original particle order."""

    def __init__(self, x, y, z, vx, vy, vz, masses, dt, theta=0.75, eps=None, mac="bh", max_leaf_n=16, ncrit=128,
                 device=0, track_integrals=False, G=1.0):
        import torch
        import rakau_amd

        self.torch = torch
        self.dev = torch.device("cuda", device)
        dtype = np.dtype(x.dtype)
        tt = torch.float32 if dtype == np.float32 else torch.float64
        up = lambda v: torch.as_tensor(np.ascontiguousarray(v), dtype=tt).to(self.dev)
        self.pos = [up(x), up(y), up(z)]
        self.vel = [up(vx), up(vy), up(vz)]
        self.m = up(masses)
        self.n = int(self.m.numel())
        self.dt = float(dt)
        self.G = float(G)
        self.q = 2 if track_integrals else 0
        self.eps2 = float(eps) ** 2 if eps is not None else 0.0
        self.mac_value = rakau_amd.mac_value_of(theta, mac, dtype)
        self.out = [torch.zeros(self.n, dtype=tt, device=self.dev) for _ in range(4 if track_integrals else 3)]
        self.ptrs = [t.data_ptr() for t in self.pos] + [self.m.data_ptr()]
        torch.cuda.synchronize(self.dev)
        self.state = rakau_amd.State.build_device(self.ptrs, self.n, dtype, max_leaf_n=max_leaf_n, ncrit=ncrit,
                                                  mac=mac, device=device)
        self.t_build = self.t_trav = 0.0
        self._accs()

    def _accs(self):
        self.state.acc_pot_device(self.q, self.mac_value, [t.data_ptr() for t in self.out], G=self.G, eps2=self.eps2,
                                  ordered=True)

    def step(self, timed=False):
        torch = self.torch
        h = 0.5 * self.dt
        for k in range(3):
            self.vel[k].add_(self.out[k], alpha=h)
            self.pos[k].add_(self.vel[k], alpha=self.dt)
        if timed:
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
        self.state.rebuild_device(self.ptrs)
        if timed:
            torch.cuda.synchronize(self.dev)
            t1 = time.perf_counter()
        self._accs()
        if timed:
            torch.cuda.synchronize(self.dev)
            t2 = time.perf_counter()
            self.t_build += t1 - t0
            self.t_trav += t2 - t1
        for k in range(3):
            self.vel[k].add_(self.out[k], alpha=h)

    def integrals(self):
        """Total energy K + W (needs track_integrals), centre of mass, its velocity. rakau's "potential" of particle i
        is its mutual potential energy -G m_i sum_j m_j / r_ij (tree.hpp:2432-2470), so W = 1/2 sum_i pot_i."""
        torch = self.torch
        f64 = torch.float64
        m = self.m.to(f64)
        kin = 0.5 * (m * sum(v.to(f64) ** 2 for v in self.vel)).sum()
        res = {"kinetic": float(kin)}
        if self.q == 2:
            w = 0.5 * self.out[3].to(f64).sum()
            res["potential"] = float(w)
            res["energy"] = float(kin + w)
        mt = m.sum()
        res["com"] = [float((m * p.to(f64)).sum() / mt) for p in self.pos]
        res["com_vel"] = [float((m * v.to(f64)).sum() / mt) for v in self.vel]
        return res


def run(nparts=1_000_000, steps=20, warmup=2, timestep=1e-4, theta=0.75, fp_type="float", mac="bh", a=1.0,
        max_leaf_n=16, ncrit=128, track_integrals=False, seed=0, device=0):
    import torch

    dtype = np.float32 if fp_type == "float" else np.float64
    x, y, z, vx, vy, vz = plummer_with_velocities(nparts, a, seed, dtype)
    n = x.size
    eps = 0.45 * n ** -0.73  # benchmark_leapfrog.cpp:221
    lf = Leapfrog(x, y, z, vx, vy, vz, np.full(n, 1.0 / n, dtype=dtype), timestep, theta, eps, mac, max_leaf_n, ncrit,
                  device, track_integrals)
    e0 = lf.integrals()
    for _ in range(warmup):
        lf.step()
    torch.cuda.synchronize()
    lf.t_build = lf.t_trav = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        lf.step(timed=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    e1 = lf.integrals()
    res = {"metric": "leapfrog steps/s (KDK, tree rebuilt every step, all arrays resident in HBM)",
           "value": steps / wall, "unit": "steps/s", "nparts": n, "steps": steps, "ms_per_step": 1e3 * wall / steps,
           "ms_rebuild": 1e3 * lf.t_build / steps, "ms_traversal": 1e3 * lf.t_trav / steps,
           "ms_integrator": 1e3 * (wall - lf.t_build - lf.t_trav) / steps, "dtype": "f32" if dtype == np.float32 else "f64",
           "theta": theta, "timestep": timestep, "eps": eps, "tree_size": lf.state.tree_size, "n_crit": lf.state.n_crit}
    if track_integrals:
        res["energy_start"] = e0["energy"]
        res["energy_end"] = e1["energy"]
        res["energy_rel_drift"] = abs(e1["energy"] - e0["energy"]) / abs(e0["energy"])
        res["virial_2K_over_W"] = -2.0 * e0["kinetic"] / e0["potential"]
    res["com_end"] = e1["com"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nparts", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--timestep", type=float, default=1e-4)
    ap.add_argument("--mac_value", type=float, default=0.75)
    ap.add_argument("--fp_type", choices=["float", "double"], default="float")
    ap.add_argument("--mac_type", choices=["bh", "bh_geom"], default="bh")
    ap.add_argument("--a", type=float, default=1.0)
    ap.add_argument("--max_leaf_n", type=int, default=16)
    ap.add_argument("--ncrit", type=int, default=128)
    ap.add_argument("--track-integrals", action="store_true")
    args = ap.parse_args()
    print(json.dumps(run(args.nparts, args.steps, args.warmup, args.timestep, args.mac_value, args.fp_type,
                         args.mac_type, args.a, args.max_leaf_n, args.ncrit, args.track_integrals)))


if __name__ == "__main__":
    main()
