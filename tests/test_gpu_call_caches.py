"""The per-state caches of the launch path -- captured hipGraph of a repeated call, launch plan of a repeated small call,
supergroup pre-pass output reused while tree and MAC value stay the same -- must never change a result: sequences of calls
that alternate MAC values, ranges, Q and outputs on ONE state are compared, bit for bit, with the same calls made once
each on fresh replicas (rk_state_clone), which have no history."""
import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import state_from_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [6000, 60000])
def test_call_sequences_with_history_equal_fresh_calls(n):
    import torch
    dev = torch.device("cuda", 0)
    m, x, y, z = oracle.plummer(n, np.float32)
    ot = oracle.Tree(x, y, z, m)
    st = state_from_oracle(ot)
    st.set_perm(ot.codes_perms()["perm"])
    cr = st.crit_ranges()
    ng = len(cr)
    cuts = [0, int(cr[ng // 3, 0]), int(cr[2 * ng // 3, 0]), n]
    mv = {t: rakau_amd.mac_value_of(t, "bh", np.float32) for t in (0.75, 0.5)}
    bufs = {k: [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(4)] for k in ("a", "b")}

    def call(state, q, theta, rng, buf, ordered=False, eps2=0.0):
        outs = bufs[buf][:rakau_amd.NRES[q]]
        for o in outs:
            o.zero_()
        state.acc_pot_device(q, mv[theta], [o.data_ptr() for o in outs], eps2=eps2, p_begin=rng[0], p_end=rng[1],
                             offset_output=True, ordered=ordered)
        torch.cuda.synchronize()
        return [o.cpu().numpy().copy() for o in outs]

    full, r1, r2 = (0, n), (cuts[0], cuts[1]), (cuts[1], cuts[3])
    seq = [(0, 0.75, full, "a"), (0, 0.75, full, "a"), (0, 0.75, full, "a"),      # direct, capture, replay
           (0, 0.5, full, "a"), (0, 0.75, full, "a"), (0, 0.75, full, "a"),        # other MAC value in between
           (0, 0.75, r1, "a"), (0, 0.75, r1, "a"), (0, 0.75, r1, "a"),             # sub-range: plan + graph
           (0, 0.75, r2, "a"), (0, 0.75, r2, "a"), (0, 0.75, r1, "a"),             # another range rebuilds the plan
           (1, 0.75, r1, "a"), (1, 0.75, r1, "a"), (2, 0.75, r1, "b"), (2, 0.75, r1, "b"),  # Q and outputs change
           (0, 0.5, r2, "a"), (0, 0.5, r2, "a"), (0, 0.75, full, "a"), (0, 0.75, full, "b"), (0, 0.75, full, "b")]
    for i, (q, theta, rng, buf) in enumerate(seq):
        for ordered in ((False, True) if i % 5 == 0 else (False,)):
            got = call(st, q, theta, rng, buf, ordered=ordered, eps2=1e-6 if q else 0.0)
            fresh = st.clone(0)
            ref = call(fresh, q, theta, rng, "b" if buf == "a" else "a", ordered=ordered, eps2=1e-6 if q else 0.0)
            fresh.close()
            for g, r in zip(got, ref):
                assert np.array_equal(g, r), (i, q, theta, rng, ordered)


def test_light_tail_plan_equals_plain_order(tmp_path):
    """The launch plans only permute the dispatch, and the one-launch kernels of small repeated calls (k_pc_any, k_list_any)
    run the code of the class kernels: the light-tail plan (limit lowered so that a 60k-particle tree takes that path), the
    heavy-first plan with every form of launch, full range, sub-ranges, Q = 0 / 2, ordered outputs, all kernel variants
    give the bits of RK_PLAN=0."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, numpy as np, torch, oracle, rakau_amd
from helpers import state_from_oracle
n = 60000
m, x, y, z = oracle.plummer(n, np.float32)
ot = oracle.Tree(x, y, z, m)
st = state_from_oracle(ot)
st.set_perm(ot.codes_perms()["perm"])
cr = st.crit_ranges()
cuts = [0, int(cr[len(cr) // 3, 0]), n]
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
res = {}
for variant in (0, 2, 3, 4):
    st.set_variant(variant)
    for q in (0, 2):
        for b, e in ((0, n), (cuts[0], cuts[1]), (cuts[1], cuts[2])):
            for ordered in (False, True):
                outs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(rakau_amd.NRES[q])]
                for rep in range(3):  # direct call, plan + capture, replay
                    st.acc_pot_device(q, mv, [o.data_ptr() for o in outs], eps2=1e-6, p_begin=b, p_end=e, ordered=ordered)
                torch.cuda.synchronize()
                res["v%d_q%d_%d_%d_%d" % (variant, q, b, e, ordered)] = np.stack([o.cpu().numpy() for o in outs])
np.savez(sys.argv[1], **res)
"""
    files = []
    # plain: no plan (full-range calls: one launch over the reversed class lists); tail: the light-tail plan; the rest: the heavy-first plan of small calls with its
    # one-launch kernels (k_pc_any up to 6000 critical nodes, k_list_any above; forced both ways, and the mixed forms).
    for name, extra in (("plain", {"RK_PLAN": "0"}), ("tail", {"RK_PLAN_MAX_GROUPS": "64"}),
                        ("heavy_first_auto", {}), ("pc_any", {"RK_ANY": "1"}), ("list_any", {"RK_ANY": "3"}),
                        ("pc_r2_list_any", {"RK_ANY": "2"}),
                        ("class_launches", {"RK_ANY": "0"}),
                        # calls without a plan: one launch over the class lists read backwards (full range), forced both ways, off
                        ("first_pc_any", {"RK_PLAN": "0", "RK_ANY": "1"}), ("first_list_any", {"RK_PLAN": "0", "RK_ANY": "3"}),
                        ("first_class_launches", {"RK_PLAN": "0", "RK_ANY": "0"}),
                        # first calls over the class lists read backwards instead of the order made on the device with the tree
                        ("first_pc_any_class_order", {"RK_PLAN": "0", "RK_ANY": "1", "RK_FIRST_ORDER": "0"}),
                        # graphs of forked sequences: parked, only the first two captured, never captured
                        ("forked_graphs_cap2", {"RK_PLAN": "0", "RK_ANY": "0", "RK_GRAPH_FORKED_MAX": "2"}),
                        ("forked_graphs_off", {"RK_PLAN_MAX_GROUPS": "64", "RK_GRAPH_FORKED_MAX": "0"}),
                        ("no_graphs", {"RK_GRAPH": "0"}), ("linear_graphs", {"RK_GRAPH_LINEAR": "1"})):
        env = dict(os.environ, RK_BACKTRACE="1", PYTHONFAULTHANDLER="1", **extra)
        env["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "tests"), env.get("PYTHONPATH", "")])
        f = str(tmp_path / (name + ".npz"))
        out = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        if out.returncode != 0:
            sys.stderr.write("env case %s:\n%s\n" % (name, out.stderr[-8000:]))  # whole native stack (RK_BACKTRACE)
        assert out.returncode == 0, name
        files.append(np.load(f))
    assert len(files[0].files) == 48
    for k in files[0].files:
        assert np.isfinite(files[0][k]).all()
        for other in files[1:]:
            assert np.array_equal(files[0][k], other[k]), k


@pytest.mark.parametrize("name,extra,nsig", [("one_launch_8", {"RK_GRAPH_LINEAR": "1"}, 8), ("forked_8", {"RK_PLAN": "0", "RK_ANY": "0"}, 8),
                                             ("forked_20_cap4", {"RK_PLAN": "0", "RK_ANY": "0", "RK_GRAPH_FORKED_MAX": "4"}, 20)])
def test_graph_cache_keeps_replay_for_recurring_signatures(name, extra, nsig):
    """tools/stress_graph_recurring.py: a caller alternating among `nsig` recurring signatures. Up to 8 (RK_GRAPH_CACHE) every
    signature is captured once and replayed ever after -- linear graphs (one-launch kernels) and forked ones (class kernels on
    side streams) alike (the linear ones only under RK_GRAPH_LINEAR=1: by default one-launch sequences are launched directly,
    which is faster); beyond, least-recently-used executables are evicted, the forked ones parked and re-targeted
    (hipGraphExecUpdate) instead of destroyed, and no more than RK_GRAPH_FORKED_MAX of them ever exist. Results bit-identical
    to the first result of every signature throughout."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the cache's own knobs are pinned: the test is about its default behaviour whatever the caller's environment says)
    env = dict(os.environ, RK_BACKTRACE="1", PYTHONFAULTHANDLER="1", RK_GRAPH="1", RK_GRAPH_CACHE="8")
    env.update(extra)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_graph_recurring.py"), "1500", str(nsig)],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0 and "graph cache stress ok" in out.stdout, name + "\n" + out.stdout[-2000:] + out.stderr[-6000:]


@pytest.mark.parametrize("timing", [True, False])
def test_alternating_streams_wait_for_each_other_not_for_the_device(timing):
    """VERDICT r04 item 5. One state, 100 calls alternating between two streams, while a THIRD stream is busy with a kernel that
    runs for about a second. Rounds 2-4 drained the whole device (hipDeviceSynchronize) at every stream change, i.e. every call
    waited for that kernel; now a call on another stream waits, on the device, for the event behind the state's previous call
    only. Checked: the 100 calls finish long before the busy stream does, and every result equals the one-stream result bit for
    bit (the calls share the state's pre-pass lists and launch plan: an ordering bug shows up as different bits). With the
    timing events off the state has no event behind its first call, so its FIRST stream change drains the device once -- the
    busy kernel is therefore started after that change in that case."""
    import time
    import torch
    m, x, y, z = oracle.plummer(60000, np.float32)
    st = rakau_amd.Octree(x, y, z, m).state()
    mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
    n = st.nparts
    ref = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    st.acc_pot_device(0, mv, [o.data_ptr() for o in ref])
    torch.cuda.synchronize()
    st.set_timing(timing)
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    busy = torch.cuda.Stream()
    outs = [[torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3)] for _ in range(2)]
    # two calls first: every stream has been seen, the plan of this signature exists
    for k in range(2):
        st.acc_pot_device(0, mv, [o.data_ptr() for o in outs[k]], stream=s[k].cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.stream(busy):
        t_busy0 = time.perf_counter()
        torch.cuda._sleep(int(2.0e9))  # ~1 s of spinning at ~2 GHz
    t0 = time.perf_counter()
    for it in range(100):
        k = it & 1
        st.acc_pot_device(0, mv, [o.data_ptr() for o in outs[k]], stream=s[k].cuda_stream)
    s[0].synchronize()
    s[1].synchronize()
    t_calls = time.perf_counter() - t0
    still_busy = not busy.query()
    busy.synchronize()
    t_busy = time.perf_counter() - t_busy0
    assert still_busy and t_calls < 0.5 * t_busy, (t_calls, t_busy, still_busy)
    for k in range(2):
        for a, b in zip(outs[k], ref):
            assert torch.equal(a, b)
