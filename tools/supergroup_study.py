"""Offline study: how much of the per-group list building could be shared by a 'supergroup' (an ancestor node
holding several critical nodes)? Uses the oracle's tree (study tool only)."""
import sys, numpy as np
sys.path.insert(0, ".")
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
smax = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m, x, y, z = oracle.plummer(n, np.float32)
t = oracle.Tree(x, y, z, m)
nd = t.nodes(); crit = t.crit_nodes()
xs, ys, zs, ms = t.parts_u()
pos = np.stack([xs, ys, zs], axis=1).astype(np.float64)
com = nd["props"][:, :3].astype(np.float64); dim2 = nd["dims"][:, 0].astype(np.float64)
nch = nd["n_children"].astype(np.int64); beg = nd["begin"].astype(np.int64); end = nd["end"].astype(np.int64)
theta = 0.75; mv = 1.0 / theta ** 2
nn = len(nch)
# supergroups: first nodes (DFS) with npart <= smax that are not inside a critical node... i.e. cover by nodes with npart<=smax or leaf
sup = []
i = 0
while i < nn:
    if end[i] - beg[i] <= smax or nch[i] == 0:
        sup.append(i); i += nch[i] + 1
    else:
        i += 1
cb = crit[:, 1].astype(np.int64)
rng = np.random.default_rng(0)
sel = rng.choice(len(sup), 60, replace=False)
tot = dict(groups=0, s_tests=0, s_accept=0, s_open=0, s_resid=0, g_tests=0, g_tests_base=0, g_accept_extra=0)
def children(i):
    c = i + 1; out = []
    while c <= i + nch[i]:
        out.append(c); c += nch[c] + 1
    return out
for si in sel:
    S = sup[si]
    P = pos[beg[S]:end[S]]; lo, hi = P.min(0), P.max(0)
    g0 = np.searchsorted(cb, beg[S]); g1 = np.searchsorted(cb, end[S])
    groups = list(range(g0, g1))
    if not groups: continue
    tot["groups"] += len(groups)
    # S-level traversal
    stack = [0]; resid = []; common = 0
    while stack:
        i = stack.pop()
        if i <= S <= i + nch[i]:   # ancestor or S itself
            if i != S: stack.extend(children(i))
            continue
        tot["s_tests"] += 1
        c = com[i]; lh = dim2[i] * mv
        dlo = np.maximum(0, np.maximum(lo - c, c - hi)); dmin2 = (dlo ** 2).sum()
        dhi = np.maximum(np.abs(lo - c), np.abs(hi - c)); dmax2 = (dhi ** 2).sum()
        if dmin2 > lh * 1.00001: common += 1; tot["s_accept"] += 1
        elif dmax2 <= lh and nch[i] > 0: tot["s_open"] += 1; stack.extend(children(i))
        else: resid.append(i); tot["s_resid"] += 1
    # per-group: baseline visits and residual visits
    for g in groups:
        b, e = int(crit[g, 1]), int(crit[g, 2]); Pg = pos[b:e]
        # find crit node idx
        def visits_from(starts):
            cnt = 0; st = list(starts)
            while st:
                i = st.pop()
                if beg[i] <= b and e <= end[i] and (end[i]-beg[i] > e-b or True) and (i <= gnode <= i + nch[i]):
                    if i != gnode: st.extend(children(i))
                    continue
                cnt += 1
                d2 = ((com[i] - Pg) ** 2).sum(1)
                if (dim2[i] * mv >= d2).any():
                    if nch[i] > 0: st.extend(children(i))
            return cnt
        # crit node index: node with begin==b,end==e, deepest first occurrence (critical)
        cand = np.where((beg == b) & (end == e))[0]
        gnode = int(cand[0])
        tot["g_tests_base"] += visits_from([0])
        tot["g_tests"] += visits_from(resid if S != gnode else [])
print(tot)
G = tot["groups"]
print("per group: baseline tests %.0f ; with supergroups: S-level %.0f/group + residual %.0f/group ; common list %.0f resid frontier %.0f per S"
      % (tot["g_tests_base"]/G, tot["s_tests"]/G, tot["g_tests"]/G, tot["s_accept"]/len(sel), tot["s_resid"]/len(sel)))
