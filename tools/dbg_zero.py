import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle, rakau_amd
rng = oracle.Rng(6)
m, x, y, z = rng.uniform_particles(4000, 1.0, np.float64)
x[:700], y[:700], z[:700] = 0.25, 0.25, -0.125
mz = np.zeros_like(m)
for builder in ("device", "host"):
    if builder == "device":
        st = rakau_amd.State.build(x, y, z, mz, box_size=1.0, mac="bh_geom")
    else:
        t = rakau_amd.Octree(x, y, z, mz, box_size=1.0, mac="bh_geom"); st = t.state()
    for variant in (1, 2):
        st.set_variant(variant)
        res = st.acc_pot(2, rakau_amd.mac_value_of(0.75, "bh_geom", np.float64), eps2=1e-4)
        for k, r in enumerate(res):
            bad = np.where(~(r == 0))[0]
            print(builder, "variant", variant, "out", k, "nbad", len(bad), r[bad[:5]], bad[:5])
