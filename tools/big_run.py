#!/usr/bin/env python3
"""A problem sized for the 288 GB of one MI355X: N particles (default 256M) of a Plummer sphere generated in HBM, tree
built on the GPU (rk_state_build_device), accs_u() at theta = 0.75, direct sums on sampled particles as the check.
big_run.py [n] [radius_clip]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 256_000_000
clip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(20261003)
t0 = time.perf_counter()
u = torch.rand(n, device=dev, generator=g, dtype=torch.float64).clamp_(1e-12, 1 - 1e-12)
r = (1.0 / torch.sqrt(u.pow(-2.0 / 3.0) - 1.0))
if clip > 0:
    r = r.clamp_(max=clip)
del u
ct = torch.rand(n, device=dev, generator=g, dtype=torch.float64) * 2 - 1
ph = torch.rand(n, device=dev, generator=g, dtype=torch.float64) * (2 * np.pi)
st = torch.sqrt(1 - ct * ct)
x = (r * st * torch.cos(ph)).float().contiguous()
y = (r * st * torch.sin(ph)).float().contiguous()
z = (r * ct).float().contiguous()
del r, ct, ph, st
m = (torch.rand(n, device=dev, generator=g, dtype=torch.float32) * 1.8 + 0.1).contiguous()
torch.cuda.synchronize()
t_gen = time.perf_counter() - t0
t0 = time.perf_counter()
state = rakau_amd.State.build_device([x.data_ptr(), y.data_ptr(), z.data_ptr(), m.data_ptr()], n, np.float32)
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
ti = state.tree_info()
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
outs = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3)]
ptrs = [o.data_ptr() for o in outs]
ms = []
for _ in range(3):
    state.acc_pot_device(0, mv, ptrs, ordered=True)
    ms.append(state.last_kernel_ms())
torch.cuda.synchronize()
# Direct sums (fp64) for sampled particles, original order thanks to the ordered outputs.
idx = torch.randint(0, n, (12,), device=dev, generator=g)
worst = 0.0
for i in idx.tolist():
    dx, dy, dz = (x - x[i]).double(), (y - y[i]).double(), (z - z[i]).double()
    r2 = dx * dx + dy * dy + dz * dz
    w = torch.where(r2 > 0, m.double() * r2.clamp_min(1e-300).pow(-1.5), torch.zeros_like(r2))
    ex = torch.stack([(dx * w).sum(), (dy * w).sum(), (dz * w).sum()])
    got = torch.stack([outs[0][i], outs[1][i], outs[2][i]]).double()
    worst = max(worst, float((got - ex).norm() / ex.norm()))
    del dx, dy, dz, r2, w
fin = all(bool(torch.isfinite(o).all()) for o in outs)
cnt = state.count_interactions(mv)
print(json.dumps({"nparts": n, "generate_s": round(t_gen, 2), "device_build_s": round(t_build, 3), "box_size": ti["box_size"],
                  "n_nodes": int(state.tree_size) if hasattr(state, "tree_size") else None, "n_crit": int(state.n_crit), "kernel_ms": [round(v, 2) for v in ms],
                  "Mparticles_per_s": round(n / (min(ms) * 1e-3) / 1e6, 1),
                  "interactions_per_particle": round((cnt["com"] + cnt["pp"] + cnt["self"]) / n, 1),
                  "finite": fin, "worst_rel_err_vs_direct_sum_12_samples": worst,
                  "hbm_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))
