"""rk_state_create(): the device buffers of a state derived from a HOST-built tree. Since round 4 the derivation runs on the
device (rk_build.hip: convert_device -- the caller's particle arrays and node records are uploaded as they are); the host loops
of round 3 remain behind RK_CREATE_ON_HOST=1. Both must give the same state: same sizes, same critical nodes, and traversal
results that agree bit for bit, for every flavour the seam instantiates. Malformed node arrays are refused, not traversed."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import state_from_oracle, oracle_nodes_aos

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = """
import sys, numpy as np, oracle, rakau_amd
from helpers import state_from_oracle
from rakau_amd import mac_value_of
res = {}
rng = oracle.Rng(3)
for dtype in (np.float32, np.float64):
    for mac in ("bh", "bh_geom"):
        m, x, y, z = oracle.plummer(40000, dtype)
        ot = oracle.Tree(x, y, z, m, mac=mac)
        st = state_from_oracle(ot)
        key = "%s_%s" % (np.dtype(dtype).name, mac)
        res[key + "_info"] = np.array([st.nparts, st.tree_size, st.n_crit])
        res[key + "_crit"] = st.crit_ranges()
        for q in (0, 2):
            got = st.acc_pot(q, mac_value_of(0.75, mac, dtype), eps2=1e-6)
            res[key + "_q%d" % q] = np.stack(got)
        cr = st.crit_ranges()
        cut = int(cr[len(cr) // 3, 0])
        res[key + "_shard"] = np.stack(st.acc_pot(0, mac_value_of(0.75, mac, dtype), p_begin=cut, p_end=st.nparts, offset_output=False))
    # quadtree, odd tree parameters, oversized critical nodes
    m, x, y = rng.uniform_particles(30000, 1.0, dtype, ndim=2)
    ot = oracle.Tree(x, y, None, m, box_size=1.0, max_leaf_n=5, ncrit=40, ndim=2)
    st = state_from_oracle(ot)
    res["quad_%s" % np.dtype(dtype).name] = np.stack(st.acc_pot(2, mac_value_of(0.6, "bh", dtype), eps2=1e-4))
    m, x, y, z = rng.uniform_particles(9000, 1.0, dtype)
    x[:1200], y[:1200], z[:1200] = 0.1, 0.2, -0.3
    ot = oracle.Tree(x, y, z, m, box_size=1.0, max_leaf_n=300, ncrit=1300)
    st = state_from_oracle(ot)
    res["big_%s" % np.dtype(dtype).name] = np.stack(st.acc_pot(2, mac_value_of(0.6, "bh", dtype), eps2=1e-4))
    res["big_%s_cls" % np.dtype(dtype).name] = st.crit_ranges()
np.savez(sys.argv[1], **res)
"""


def test_device_and_host_conversion_give_the_same_state(tmp_path):
    files = []
    for name, extra in (("device", {}), ("host", {"RK_CREATE_ON_HOST": "1"})):
        env = dict(os.environ, RK_BACKTRACE="1", **extra)
        env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
        f = str(tmp_path / (name + ".npz"))
        out = subprocess.run([sys.executable, "-c", CODE, f], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, name + "\n" + out.stderr[-4000:]
        files.append(np.load(f))
    assert len(files[0].files) == 26
    for k in files[0].files:
        assert np.isfinite(files[0][k]).all()
        assert np.array_equal(files[0][k], files[1][k]), k


def test_malformed_node_arrays_are_refused():
    m, x, y, z = oracle.plummer(5000, np.float32)
    ot = oracle.Tree(x, y, z, m)
    p = ot.parts_u()
    good = oracle_nodes_aos(ot)
    rakau_amd.State(p[0], p[1], p[2], p[3], good, ncrit=ot.ncrit).close()

    def broken(edit):
        a = good.copy()
        edit(a)
        return a

    cases = [
        (lambda a: a["end"].__setitem__(7, a["begin"][7]), "inconsistent tree node at index 7"),            # empty range
        (lambda a: a["end"].__setitem__(3, 5001), "inconsistent tree node at index 3"),                      # past the particles
        (lambda a: a["n_children"].__setitem__(10, len(a)), "inconsistent tree node at index 10"),           # more descendants than nodes
        (lambda a: a["n_children"].__setitem__(0, a["n_children"][0] - 1), "inconsistent"),                  # the root loses its last node
    ]
    for edit, msg in cases:
        with pytest.raises(ValueError, match=msg):
            rakau_amd.State(p[0], p[1], p[2], p[3], broken(edit), ncrit=ot.ncrit)
    # a state is still fine afterwards
    st = rakau_amd.State(p[0], p[1], p[2], p[3], good, ncrit=ot.ncrit)
    assert st.n_crit == ot.n_crit


ORPHAN_CODE = """
import sys, numpy as np, oracle, rakau_amd
from helpers import oracle_nodes_aos
m, x, y, z = oracle.plummer(5000, np.float32)
ot = oracle.Tree(x, y, z, m)
p = ot.parts_u()
good = oracle_nodes_aos(ot)
n = len(good)
# an internal node in the middle of the array whose children are internal too
mid = [k for k in range(n // 3, n) if good["n_children"][k] > 20][0]
first_child_span = int(good["n_children"][mid + 1]) + 1
def edits():
    yield "root without children", lambda a: a["n_children"].__setitem__(0, 0)
    yield "root count truncated to its first child", lambda a: a["n_children"].__setitem__(0, int(a["n_children"][1]) + 1)
    yield "inner node keeps its first child only", lambda a: a["n_children"].__setitem__(mid, first_child_span)
    yield "inner node becomes a leaf", lambda a: a["n_children"].__setitem__(mid, 0)
for rep in range(3):   # warm and cold block cache alike
    for name, edit in edits():
        a = good.copy()
        edit(a)
        try:
            rakau_amd.State(p[0], p[1], p[2], p[3], a, ncrit=ot.ncrit)
        except ValueError as e:
            # (the two conversions word some of these differently: "inconsistent ..." / "tree node N has more than 8 children")
            assert "inconsistent" in str(e) or "children" in str(e), (name, str(e))
        else:
            raise SystemExit("accepted: " + name)
    st = rakau_amd.State(p[0], p[1], p[2], p[3], good, ncrit=ot.ncrit)
    assert st.n_crit == ot.n_crit
    st.close()
print("ORPHANS_REFUSED")
"""


@pytest.mark.parametrize("where", ["device", "host"])
def test_nodes_no_parent_claims_are_refused_under_a_poisoned_pool(where):
    """node arrays in which some node is claimed by no parent (a root with n_children = 0 or truncated, an inner
    node whose count covers only part of its subtree). The device conversion keeps parent[] in a block from the recycling
    pool; with RK_POOL_POISON=255 every block is handed out filled with 0xff, so a read of a parent index nobody wrote is
    an out-of-range index, deterministically. Both conversions must refuse such trees with the reference-style error."""
    env = dict(os.environ, RK_POOL_POISON="255", RK_BACKTRACE="1")
    if where == "host":
        env["RK_CREATE_ON_HOST"] = "1"
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", ORPHAN_CODE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "ORPHANS_REFUSED" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_state_create_time_4m():
    """Not a parity test: prints what rk_state_create() costs at 4M either way (run with -s)."""
    code = """
import sys, time, numpy as np, rakau_amd
from bench import plummer_numpy
m, x, y, z = plummer_numpy(4_000_000, "float32")
t = rakau_amd.Octree(x, y, z, m)
p = t.p_its_u(); nodes = t.nodes()
ts = []
for i in range(5):
    t0 = time.perf_counter()
    s = rakau_amd.State(p[0], p[1], p[2], p[3], nodes, ncrit=128)
    ts.append(time.perf_counter() - t0)
    s.close()
print("CREATE_MS " + " ".join("%.1f" % (v * 1e3) for v in ts))
"""
    for name, extra in (("device", {}), ("host", {"RK_CREATE_ON_HOST": "1"})):
        env = dict(os.environ, **extra)
        env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        print("\nrk_state_create at 4M fp32, conversion on the %s: %s" % (name, [l for l in out.stdout.splitlines() if l.startswith("CREATE_MS")][0]))
