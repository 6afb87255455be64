#!/usr/bin/env python3
"""Dynamic event counts of one full-range accs_u() call (library built with -DRK_COUNTS: tools/build_variant_full.sh counts
-DRK_COUNTS; RAKAU_AMD_LIB=rakau_amd/lib_counts/librakau_amd.so): the library prints the counters of the PREVIOUS call on
stderr at every call; this script makes two calls and labels the second printout.  usage: counts_probe.py [nparts]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "nodes", 1: "batches", 2: "candidates in batches", 3: "exact batches (lane = candidate)", 4: "  sum of their target counts",
         5: "exact batches (lane = target)", 6: "  sum of their candidates", 7: "undecided candidates", 8: "leaf-drain rounds",
         9: "leaf-gather steps (8 particles per lane)", 10: "leaves gathered", 11: "leaf particles gathered", 12: "tiles evaluated",
         13: "dense-loop trips (sum of full)", 14: "remainder steps", 15: "sources in tiles", 16: "common-list copy steps",
         17: "common sources", 18: "own-particle tiles", 19: "  their dense-loop trips", 20: "sum NS", 21: "sum TP*NS (lanes on)",
         22: "sum T*NS", 23: "accepted nodes", 24: "opened internal nodes", 25: "  candidates in lane = candidate exact batches",
         26: "sum T", 27: "own-particle remainder steps"}
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, rakau_amd
    from bench import plummer_numpy
    n = int(float(sys.argv[2]))
    m, x, y, z = plummer_numpy(n, "float32")
    st = rakau_amd.Octree(x, y, z, m).state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    for _ in range(3):
        st.acc_pot_device(0, mv, [o.data_ptr() for o in outs])
        torch.cuda.synchronize()
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "4000000"
out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], capture_output=True, text=True,
                     env=dict(os.environ, RK_GRAPH="0", RK_SUPER_CACHE="0"))
blocks = [l for l in out.stderr.splitlines() if l.startswith("RK_COUNTS prev")]
if len(blocks) < 12:
    print(out.stderr[-3000:]); sys.exit(1)
last = blocks[-4:]  # counters of the second call, printed at the third
rows = [[int(v) for v in l.split(":")[1].split()] for l in last]
print("event counts of one accs_u() call, N = %s, per lane-mapping class R = 1..4 and in all" % n)
for i in range(28):
    print("%-48s %14d %14d %14d %14d | %15d" % (NAMES.get(i, str(i)), rows[0][i], rows[1][i], rows[2][i], rows[3][i], sum(r[i] for r in rows)))
