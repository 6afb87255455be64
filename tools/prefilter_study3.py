"""Offline study (oracle tree): which TWO probe targets leave the fewest candidates undecided after the bounding-box
accept test? (The kernel's probes are the first and last particle of the group in Morton order.)"""
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
sel = rng.choice(len(crit), 200, replace=False)
names = ["first_last", "axis_extremes", "farthest_pair", "corner_nearest", "three_axis", "first_last+axis"]
und = dict.fromkeys(names, 0); visits = 0; notbox = 0
for g in sel:
    ccode, b, e = (int(v) for v in crit[g])
    P = pos[b:e]; T = len(P)
    lo, hi = P.min(0), P.max(0)
    ax = int(np.argmax(hi - lo))
    pa = [int(P[:, ax].argmin()), int(P[:, ax].argmax())]
    D = ((P[:, None, :] - P[None, :, :]) ** 2).sum(2); fp = list(np.unravel_index(D.argmax(), D.shape))
    cn = [int(((P - lo) ** 2).sum(1).argmin()), int(((P - hi) ** 2).sum(1).argmin())]
    ax3 = list({int(P[:, k].argmin()) for k in range(3)} | {int(P[:, k].argmax()) for k in range(3)})
    sets = {"first_last": [0, T - 1], "axis_extremes": pa, "farthest_pair": fp, "corner_nearest": cn, "three_axis": ax3,
            "first_last+axis": [0, T - 1] + pa}
    clevel = (ccode.bit_length() - 1) // 3
    i = 0; nn = len(nch)
    while i < nn:
        sl = int(level[i])
        if sl <= clevel and (ccode >> (3 * (clevel - sl))) == int(code[i]):
            i += 1 + (nch[i] if int(code[i]) == ccode else 0); continue
        c = com[i]; lh = dim2[i] * mv
        d2 = ((c - P) ** 2).sum(1); fail = bool((lh >= d2).any())
        visits += 1
        dlo = np.maximum(0, np.maximum(lo - c, c - hi))
        if not ((dlo ** 2).sum() > lh * 1.00001):
            notbox += 1
            for k, idx in sets.items():
                if not (lh >= d2[idx]).any():
                    und[k] += 1
        i += 1 if fail else nch[i] + 1
print("visits", visits, "not accepted by the box", notbox)
for k in names:
    print("%-18s undecided %.4f of visits" % (k, und[k] / visits))
