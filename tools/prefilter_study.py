"""Offline study: how many MAC decisions of the traversal can be taken from the group's bounding box
(without looping over all targets)? Uses the oracle's tree (test infrastructure; study tool only)."""
import sys, numpy as np
sys.path.insert(0, ".")
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m, x, y, z = oracle.plummer(n, np.float32)
t = oracle.Tree(x, y, z, m)
nd = t.nodes(); crit = t.crit_nodes()
xs, ys, zs, ms = t.parts_u()
pos = np.stack([xs, ys, zs], axis=1).astype(np.float64)
com = nd["props"][:, :3].astype(np.float64); dim2 = nd["dims"][:, 0].astype(np.float64)
nch = nd["n_children"].astype(np.int64); code = nd["code"]; level = nd["level"].astype(np.int64)
theta = 0.75; mv = 1.0 / theta ** 2
rng = np.random.default_rng(0)
sel = rng.choice(len(crit), 150, replace=False)
tot = dict(visits=0, acc_box=0, rej_probe=0, undecided=0, acc_sphere=0, rej_any2=0)
for g in sel:
    ccode, b, e = (int(v) for v in crit[g])
    P = pos[b:e]
    lo, hi = P.min(0), P.max(0); ctr = 0.5 * (lo + hi); rad = np.sqrt(((P - ctr) ** 2).sum(1).max())
    clevel = (ccode.bit_length() - 1) // 3
    i = 0; nn = len(nch)
    while i < nn:
        sl = int(level[i])
        if sl <= clevel and (ccode >> (3 * (clevel - sl))) == int(code[i]):
            i += 1 + (nch[i] if int(code[i]) == ccode else 0); continue
        c = com[i]; lh = dim2[i] * mv
        d2 = ((c - P) ** 2).sum(1); fail = (lh >= d2).any()
        tot["visits"] += 1
        # box lower bound on distance
        dlo = np.maximum(0, np.maximum(lo - c, c - hi)); dmin2 = (dlo ** 2).sum()
        dc = np.sqrt(((c - ctr) ** 2).sum())
        sph_acc = (dc - rad) > 0 and (dc - rad) ** 2 > lh * 1.00001
        box_acc = dmin2 > lh * 1.00001
        probe_rej = lh >= d2[0] or lh >= d2[-1]
        if box_acc: tot["acc_box"] += 1
        if sph_acc: tot["acc_sphere"] += 1
        if probe_rej: tot["rej_probe"] += 1
        # nearest-corner probe: target closest to c along... use target nearest to ctr-projected? cheap alt: 2 probes
        if not box_acc and not probe_rej: tot["undecided"] += 1
        assert not (box_acc and fail)
        i += 1 if fail else nch[i] + 1
print(tot, {k: round(v / tot["visits"], 3) for k, v in tot.items()})
