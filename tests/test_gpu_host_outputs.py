"""rk_acc_pot() -- the seam's own signature, host output arrays (rocm_state::acc_pot of the reference,
detail/rocm_fwd.hpp:38-40) -- delivers the same bits whichever way the results travel: small results through a device
buffer, large results through the pinned staging buffer and the delivery threads, and output arrays that are themselves pinned (rk_host_alloc, torch pinned tensors), which the kernels write
directly. Checked against rk_acc_pot_device() + an explicit copy, for compact and offset outputs, unaligned ranges."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import rakau_amd
from helpers import state_from_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_result(st, q, mv, n, dtype, b, e, eps2):
    import torch
    outs = [torch.zeros(n, dtype=getattr(torch, np.dtype(dtype).name), device="cuda") for _ in range(rakau_amd.NRES[q])]
    st.acc_pot_device(q, mv, [o.data_ptr() for o in outs], eps2=eps2, p_begin=b, p_end=e, offset_output=True)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs]


# (1.5M fp32: results of 18-24 MB, i.e. the two-part staged call whose first part is delivered while the second is traversed)
@pytest.mark.parametrize("dtype,n", [(np.float32, 400000), (np.float64, 150000), (np.float32, 20000), (np.float32, 1500000)])
def test_pageable_pinned_and_device_outputs_agree(dtype, n):
    import torch
    m, x, y, z = oracle.plummer(n, dtype)
    ot = oracle.Tree(x, y, z, m)
    st = state_from_oracle(ot)
    mv = rakau_amd.mac_value_of(0.75, "bh", dtype)
    cr = st.crit_ranges()
    ng = len(cr)
    for q in (0, 2):
        nres = rakau_amd.NRES[q]
        for b, e in ((0, n), (int(cr[ng // 7, 0]), int(cr[(6 * ng) // 7, 0]))):
            ref = device_result(st, q, mv, n, dtype, b, e, 1e-6)
            # pageable arrays, offset and compact
            out = st.acc_pot(q, mv, eps2=1e-6, p_begin=b, p_end=e, out=[np.zeros(n, dtype=dtype) for _ in range(nres)])
            for k in range(nres):
                assert np.array_equal(out[k][b:e], ref[k][b:e])
                assert not out[k][:b].any() and not out[k][e:].any()
            out = st.acc_pot(q, mv, eps2=1e-6, p_begin=b, p_end=e, offset_output=False,
                             out=[np.zeros(e - b + 3, dtype=dtype)[3:] for _ in range(nres)])  # odd alignment
            for k in range(nres):
                assert np.array_equal(out[k], ref[k][b:e])
            # pinned arrays from rk_host_alloc(): written by the kernels themselves
            pin = [rakau_amd.pinned_empty(n, dtype) for _ in range(nres)]
            for p in pin:
                p[:] = 0
            st.acc_pot(q, mv, eps2=1e-6, p_begin=b, p_end=e, out=pin)
            for k in range(nres):
                assert np.array_equal(pin[k][b:e], ref[k][b:e])
                assert not pin[k][:b].any() and not pin[k][e:].any()
            # torch's pinned tensors are the same kind of memory
            tp = [torch.zeros(e - b, dtype=getattr(torch, np.dtype(dtype).name)).pin_memory() for _ in range(nres)]
            st.acc_pot(q, mv, eps2=1e-6, p_begin=b, p_end=e, offset_output=False, out=[t.numpy() for t in tp])
            for k in range(nres):
                assert np.array_equal(tp[k].numpy(), ref[k][b:e])
            # a mix of pinned and pageable arrays takes the staging path
            mix = [pin[0]] + [np.zeros(n, dtype=dtype) for _ in range(nres - 1)]
            mix[0][:] = 0
            st.acc_pot(q, mv, eps2=1e-6, p_begin=b, p_end=e, out=mix)
            for k in range(nres):
                assert np.array_equal(mix[k][b:e], ref[k][b:e])
    st.close()


def test_pinned_block_lifetime_and_errors():
    a = rakau_amd.pinned_empty(1000, np.float32)
    a[:] = 7
    b = a[10:20]
    del a
    assert float(b.sum()) == 70.0
    z = rakau_amd.pinned_empty(0, np.float64)
    assert z.size == 0
    import ctypes as C
    from rakau_amd import _capi
    assert _capi.lib().rk_host_alloc(None, 16) != 0
    assert b"null" in _capi.lib().rk_last_error()
    assert _capi.lib().rk_host_free(None) == 0


@pytest.mark.parametrize("env", [{"RK_HOST_THREADS": "1"}, {"RK_HOST_THREADS": "3"}, {"RK_HOST_REGISTER": "1"}])
def test_delivery_knobs_do_not_change_results(env):
    code = """
import numpy as np, oracle, rakau_amd
from helpers import state_from_oracle
n = 300000
m, x, y, z = oracle.plummer(n, np.float32)
st = state_from_oracle(oracle.Tree(x, y, z, m))
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
a = st.acc_pot(0, mv)
p = [rakau_amd.pinned_empty(n, np.float32) for _ in range(3)]
st.acc_pot(0, mv, out=p)
print("SUM", repr(float(np.sum([np.abs(v).sum(dtype=np.float64) for v in a]))), all(np.array_equal(u, v) for u, v in zip(a, p)))
"""
    def run(extra):
        e = dict(os.environ, **extra)
        e["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), e.get("PYTHONPATH", "")])
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        return [l for l in out.stdout.splitlines() if l.startswith("SUM")][0]
    base = run({})
    assert base.endswith("True")
    assert run(env) == base


@pytest.mark.parametrize("register", ["0", "1"])
def test_output_arrays_that_are_slices_of_one_allocation(register):
    """The layout of benchmark/benchmark_acc.cpp: one buffer of N-element blocks, so the output arrays share pages. Results
    equal the device-output call, bytes around the slices are untouched, repeated calls and a call into other arrays work --
    through the staging buffer (default) and with the opt-in scoped registration (RK_HOST_REGISTER=1: the slices must travel
    as ONE registered range). The registration runs in a fresh process whose only pageable HIP copies come from arrays that
    stay alive: it is unsafe next to other pinnings of the same pages, hence opt-in."""
    code = """
import numpy as np, torch, oracle, rakau_amd
from helpers import state_from_oracle
n = 250000
m, x, y, z = oracle.plummer(n, np.float32)
st = state_from_oracle(oracle.Tree(x, y, z, m))
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
d = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(4)]
st.acc_pot_device(2, mv, [t.data_ptr() for t in d])
torch.cuda.synchronize()
ref = [t.cpu().numpy() for t in d]
buf = np.full(4 * n + 7, -7.0, dtype=np.float32)
outs = [buf[5 + k * n:5 + (k + 1) * n] for k in range(4)]  # back to back, unaligned
for _ in range(3):
    st.acc_pot(2, mv, out=outs)
    for o, r in zip(outs, ref):
        assert np.array_equal(o, r)
    assert np.all(buf[:5] == -7.0) and np.all(buf[5 + 4 * n:] == -7.0)
other = st.acc_pot(2, mv)
for o, r in zip(other, ref):
    assert np.array_equal(o, r)
print("slices ok")
"""
    e = dict(os.environ, RK_HOST_REGISTER=register)
    e["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), e.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert out.returncode == 0 and "slices ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("dtype,n", [(np.float32, 20000), (np.float32, 700000), (np.float64, 150000)])
def test_ordered_host_outputs(dtype, n):
    """rk_acc_pot(RK_OUT_OFFSET | RK_OUT_ORDERED): accs_o / pots_o / accs_pots_o into host arrays (tree.hpp:3320-3330). The
    kernels scatter through perm into HBM, the ordered arrays travel whole: same bits as the Morton-order result indexed by
    the inverse permutation, for pageable (small: straight copy; large: staged, delivered by the host threads) and pinned
    arrays. Sub-ranges and a state without perm are refused."""
    m, x, y, z = oracle.plummer(n, dtype)
    ot = oracle.Tree(x, y, z, m)
    st = state_from_oracle(ot)
    mv = rakau_amd.mac_value_of(0.75, "bh", dtype)
    with pytest.raises(ValueError):
        st.acc_pot(0, mv, ordered=True)  # no permutation yet
    perm = ot.codes_perms()["perm"]
    st.set_perm(perm)
    for q in (0, 1, 2):
        nres = rakau_amd.NRES[q]
        ref = st.acc_pot(q, mv, eps2=1e-6)
        for rep in range(2):
            out = st.acc_pot(q, mv, eps2=1e-6, ordered=True, out=[np.full(n + 1, 7, dtype=dtype)[1:] for _ in range(nres)])
            for k in range(nres):
                assert np.array_equal(out[k][perm], ref[k])
        pin = [rakau_amd.pinned_empty(n, dtype) for _ in range(nres)]
        st.acc_pot(q, mv, eps2=1e-6, ordered=True, out=pin)
        for k in range(nres):
            assert np.array_equal(pin[k][perm], ref[k])
    cr = st.crit_ranges()
    with pytest.raises(ValueError):
        st.acc_pot(0, mv, ordered=True, p_begin=int(cr[1, 0]))
    # a compact call after the ordered ones is untouched by them
    again = st.acc_pot(0, mv, eps2=1e-6)
    ref0 = st.acc_pot(0, mv, eps2=1e-6)
    assert all(np.array_equal(a, b) for a, b in zip(again, ref0))
