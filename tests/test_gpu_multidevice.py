"""The multi-device host leg (kwargs::split = {cpu, dev0, dev1, ...}, tree.hpp:3150-3240 of the reference; one state and
one host thread per device, replication with rk_state_clone, cuts at critical nodes) on a 1-GPU box: RK_ALIAS_DEVICES=4
makes the library report four LOGICAL devices mapped onto the one GPU, so everything but the physical xGMI transport runs."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import oracle, rakau_amd
    from rakau_amd import _capi
    assert _capi.lib().rk_device_count() == 4
    m, x, y, z = oracle.plummer(60000, np.float32)
    t = rakau_amd.Octree(x, y, z, m)
    base = t.accs_pots_u(0.75, eps=1e-3, G=2.0)
    cpu = t.cpu_acc_pot_u(2, 0.75, eps=1e-3, G=2.0)
    crit = t.crit_nodes()[:, 1].astype(np.int64)
    def snap(frac):
        i = int(np.searchsorted(crit, int(frac * t.nparts), side="left"))
        return int(crit[i]) if i < len(crit) else t.nparts
    for split in ([0.1, 0.3, 0.3, 0.3], [0.0, 1.0, 3.0, 1.0, 2.0], [2.0, 1.0, 1.0]):
        got = t.accs_pots_u(0.75, eps=1e-3, G=2.0, split=split)
        cut = snap(split[0] / sum(split))
        for g, b, c in zip(got, base, cpu):
            assert np.array_equal(g[cut:], b[cut:]), split   # every device reproduces the single-device result bit for bit
            assert np.array_equal(g[:cut], c[:cut]), split   # the CPU share is the CPU engine's
    # A device share below rk_min_size() sends the whole call to the CPU engine (tree.hpp:3191-3199 of the reference).
    for g, c in zip(t.accs_pots_u(0.75, eps=1e-3, G=2.0, split=[0.0, 1.0, 1.0, 0.0, 2.0]), cpu):
        assert np.array_equal(g, c)
    # _o outputs and a second call (replicas are reused).
    perm = t.perm().astype(np.int64)
    for u, o in zip(t.accs_u(0.75, split=[0, 1, 1, 1, 1]), t.accs_o(0.75, split=[0, 1, 1, 1, 1])):
        assert np.array_equal(o[perm], u)
    try:
        t.accs_u(0.75, split=[1.0] * 6)
        raise SystemExit("no error for 5 accelerators")
    except ValueError as e:
        assert "refers to 5 accelerators, but only 4 were detected" in str(e), e
    # rk_state_clone directly: a replica on logical device 3 (+ ordered output through the replicated permutation).
    st = t.state()
    st.set_perm(t.perm())
    rep = st.clone(3)
    assert rep.device == 3 and (rep.nparts, rep.tree_size, rep.n_crit) == (st.nparts, st.tree_size, st.n_crit)
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    for a, b in zip(st.acc_pot(2, mv, eps2=1e-6), rep.acc_pot(2, mv, eps2=1e-6)):
        assert np.array_equal(a, b)
    # update_particles drops every replica; the next split call rebuilds them.
    t.update_particles_u(lambda a: a[0].__imul__(np.float32(1.001)))
    b2 = t.accs_u(0.75)
    for g, b in zip(t.accs_u(0.75, split=[0, 1, 1, 1, 1]), b2):
        assert np.array_equal(g, b)
    print("multi-device ok")
""") % ROOT


def test_split_over_four_logical_devices():
    env = dict(os.environ, RK_ALIAS_DEVICES="4")
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "multi-device ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
