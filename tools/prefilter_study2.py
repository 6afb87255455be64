"""Offline study (oracle tree): what happens to the candidates the box/probe prefilter leaves undecided, and which
extra cheap test would decide most of them? For each visited (group, node): truth = any target fails the MAC."""
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
keys = ["visits", "undecided", "und_accept", "und_open", "sub2_accept", "sub4_accept", "probe4_open", "probe8_open",
        "ext6_open", "sub2_or_ext6", "sub4_or_probe8"]
tot = dict.fromkeys(keys, 0)

def boxdist2(lo, hi, c):
    d = np.maximum(0, np.maximum(lo - c, c - hi)); return (d ** 2).sum()

for g in sel:
    ccode, b, e = (int(v) for v in crit[g])
    P = pos[b:e]; T = len(P)
    lo, hi = P.min(0), P.max(0)
    clevel = (ccode.bit_length() - 1) // 3
    halves = [P[:max(1, T // 2)], P[T // 2:]] if T > 1 else [P]
    quarters = [P[k * T // 4:max(k * T // 4 + 1, (k + 1) * T // 4)] for k in range(4)] if T >= 4 else halves
    hb = [(q.min(0), q.max(0)) for q in halves]; qb = [(q.min(0), q.max(0)) for q in quarters]
    p4 = P[np.unique(np.linspace(0, T - 1, 4).astype(int))]; p8 = P[np.unique(np.linspace(0, T - 1, 8).astype(int))]
    ext = P[np.unique(np.concatenate([P.argmin(0), P.argmax(0)]))]
    i = 0; nn = len(nch)
    while i < nn:
        sl = int(level[i])
        if sl <= clevel and (ccode >> (3 * (clevel - sl))) == int(code[i]):
            i += 1 + (nch[i] if int(code[i]) == ccode else 0); continue
        c = com[i]; lh = dim2[i] * mv
        d2 = ((c - P) ** 2).sum(1); fail = bool((lh >= d2).any())
        tot["visits"] += 1
        box_acc = boxdist2(lo, hi, c) > lh * 1.00001
        probe_rej = lh >= d2[0] or lh >= d2[-1]
        if not box_acc and not probe_rej:
            tot["undecided"] += 1
            tot["und_open" if fail else "und_accept"] += 1
            s2 = min(boxdist2(l, h, c) for l, h in hb) > lh * 1.00001
            s4 = min(boxdist2(l, h, c) for l, h in qb) > lh * 1.00001
            o4 = bool((lh >= ((c - p4) ** 2).sum(1)).any()); o8 = bool((lh >= ((c - p8) ** 2).sum(1)).any())
            oe = bool((lh >= ((c - ext) ** 2).sum(1)).any())
            tot["sub2_accept"] += s2; tot["sub4_accept"] += s4; tot["probe4_open"] += o4; tot["probe8_open"] += o8
            tot["ext6_open"] += oe; tot["sub2_or_ext6"] += (s2 or oe); tot["sub4_or_probe8"] += (s4 or o8)
            assert not (s4 and fail) and not (s2 and fail)
        i += 1 if fail else nch[i] + 1
print(tot)
u = tot["undecided"]
print("undecided %.3f of visits; of those: accept %.2f open %.2f | decided by sub2 %.2f sub4 %.2f probe4 %.2f probe8 %.2f ext6 %.2f sub2|ext6 %.2f sub4|probe8 %.2f"
      % (u / tot["visits"], tot["und_accept"] / u, tot["und_open"] / u, tot["sub2_accept"] / u, tot["sub4_accept"] / u,
         tot["probe4_open"] / u, tot["probe8_open"] / u, tot["ext6_open"] / u, tot["sub2_or_ext6"] / u, tot["sub4_or_probe8"] / u))
