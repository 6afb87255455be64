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


FANOUT = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import oracle, rakau_amd
    from rakau_amd import _capi
    assert _capi.lib().rk_device_count() == 8
    m, x, y, z = oracle.plummer(40000, np.float32)
    # A tree built on the GPU whose FIRST call is a multi-device split: device 0's state has no host mirrors yet, the
    # replicas must exist (and the mirrors be filled) before the device threads start.
    t = rakau_amd.Octree(x, y, z, m, builder="device")
    got = t.accs_pots_u(0.75, eps=1e-3, split=[0, 1, 1, 1, 1, 1, 1, 1, 1])
    base = t.accs_pots_u(0.75, eps=1e-3)
    for g, b in zip(got, base):
        assert np.array_equal(g, b)
    # rk_state_clone_all directly: seven replicas, every one traverses like the original.
    st = t.state()
    reps = st.clone_all([1, 2, 3, 4, 5, 6, 7])
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    ref = st.acc_pot(0, mv)
    for d, r in enumerate(reps, 1):
        assert r.device == d and (r.nparts, r.tree_size, r.n_crit) == (st.nparts, st.tree_size, st.n_crit)
        for a, b in zip(ref, r.acc_pot(0, mv)):
            assert np.array_equal(a, b)
    print("fan-out ok")
""") % ROOT


def test_replicas_fan_out_as_a_doubling_tree():
    """rk_state_clone_all on eight logical devices: 7 replicas in ceil(log2(8)) = 3 rounds, the transfers of a round issued
    together (RK_CLONE_TRACE prints them), and a device-built tree whose first call is a split over all devices."""
    env = dict(os.environ, RK_ALIAS_DEVICES="8", RK_CLONE_TRACE="1")
    out = subprocess.run([sys.executable, "-c", FANOUT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "fan-out ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    rounds = {}
    for line in out.stderr.splitlines():
        if line.startswith("rk_state_clone_all round"):
            w = line.split()
            rounds.setdefault(int(w[2].rstrip(":")), []).append((int(w[4]), int(w[7])))
    # The explicit clone_all call is the last one traced: 1, 2 and 4 transfers in rounds 0, 1, 2.
    last = {r: v[-(1 << r):] for r, v in rounds.items()}
    assert sorted(last) == [0, 1, 2] and [len(last[r]) for r in (0, 1, 2)] == [1, 2, 4], rounds
    senders = {0}
    for r in (0, 1, 2):
        assert all(s in senders for s, _ in last[r]), (r, last[r])
        senders |= {d for _, d in last[r]}
    assert senders == set(range(8))


RCCL = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import oracle, rakau_amd
    from rakau_amd.state import Comm, State
    m, x, y, z = oracle.plummer(30000, np.float32)
    t = rakau_amd.Octree(x, y, z, m)
    st = t.state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    before = st.acc_pot(0, mv)
    comm = Comm(1, Comm.unique_id(), 0, 0)      # ncclCommInitRank, one rank
    same = State.broadcast(st, 0, 0, 0, comm)   # ncclBroadcast of the meta block and of every buffer, in place on the root
    assert same is st
    for a, b in zip(before, st.acc_pot(0, mv)):
        assert np.array_equal(a, b)
    comm.close()
    print("rccl ok")
""") % ROOT


def test_rccl_broadcast_entry_world_size_one():
    """The library's own replicate step (rk_comm_* + rk_state_broadcast: RCCL bound at run time) on a communicator of one
    rank: every ncclBroadcast call of the N-rank path executes (in place on the root) and the state is untouched."""
    out = subprocess.run([sys.executable, "-c", RCCL], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
