#!/usr/bin/env python3
"""Split traversal (variant 4) against the fused list kernel (variant 2): agreement to rounding on small trees for every
flavour, determinism, sub-range union; then kernel times over sizes for both. Uses the product only."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rakau_amd
from bench import plummer_numpy

def relvec(a, b, nd):
    a = np.stack([np.asarray(v, np.float64) for v in a[:nd]], 1); b = np.stack([np.asarray(v, np.float64) for v in b[:nd]], 1)
    den = np.linalg.norm(b, axis=1); den[den == 0] = 1
    return np.linalg.norm(a - b, axis=1) / den

ok = True
if "--skip-check" not in sys.argv:
    for dtype in ("float32", "float64"):
        for mac in ("bh", "bh_geom"):
            m, x, y, z = plummer_numpy(30000, dtype)
            t = rakau_amd.Octree(x, y, z, m, mac=mac)
            st = t.state()
            cr0 = st.crit_ranges()
            for theta in (0.75, 0.4):
                mv = rakau_amd.mac_value_of(theta, mac, np.dtype(dtype).type)
                for q in (0, 1, 2):
                    st.set_variant(2)
                    ref = st.acc_pot(q, mv, eps2=1e-6, G=1.5)
                    st.set_variant(4)
                    os.environ["RK_SL_PARTS_BELOW"] = "0"     # one wavefront per node
                    got = st.acc_pot(q, mv, eps2=1e-6, G=1.5)
                    os.environ["RK_SL_PARTS_BELOW"] = "1000000"  # one wavefront per part + k_combine
                    again = st.acc_pot(q, mv, eps2=1e-6, G=1.5)
                    det = all(np.array_equal(a, b) for a, b in zip(got, again))
                    os.environ["RK_SL_PARTS_BELOW"] = str(len(cr0) // 2)  # shards in the other form than the full range

                    cr = st.crit_ranges(); cut = int(cr[len(cr) // 3, 0])
                    lo = st.acc_pot(q, mv, eps2=1e-6, G=1.5, p_begin=0, p_end=cut, offset_output=False)
                    hi = st.acc_pot(q, mv, eps2=1e-6, G=1.5, p_begin=cut, p_end=len(x), offset_output=False)
                    uni = all(np.array_equal(a, np.concatenate([l, h])) for a, l, h in zip(got, lo, hi))
                    if q in (0, 2):
                        e = relvec(got, ref, 3).max()
                    else:
                        e = (np.abs(np.asarray(got[0], np.float64) - ref[0]) / np.abs(ref[0])).max()
                    tol = 2e-5 if dtype == "float32" else 1e-12
                    good = det and uni and e < tol and all(np.all(np.isfinite(g)) for g in got)
                    ok &= good
                    print("%s %s theta %.2f q%d: max rel diff vs fused %.2e det %s union %s %s" % (dtype, mac, theta, q, e, det, uni, "ok" if good else "FAIL"), flush=True)
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    os.environ.pop("RK_SL_PARTS_BELOW", None)

sizes = [int(float(v)) for v in (os.environ.get("SIZES", "1e5,3.5e5,1e6,4e6")).split(",")]
for n in sizes:
    m, x, y, z = plummer_numpy(n, "float32")
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ptrs = [o.data_ptr() for o in outs]
    for _ in range(30 if n >= 1000000 else 60):
        st.acc_pot_device(0, mv, ptrs)
    torch.cuda.synchronize()
    for variant in (2, 3, 4, 2, 4):
        st.set_variant(variant)
        ms = []
        for _ in range(24):
            st.acc_pot_device(0, mv, ptrs)
            ms.append(st.last_kernel_ms())
        torch.cuda.synchronize()
        print("n=%d variant %d kernel ms: median %.4f min %.4f" % (n, variant, float(np.median(ms[6:])), min(ms[6:])), flush=True)
    del st, t, outs
