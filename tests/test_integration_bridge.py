"""integration/rakau_amd_bridge.cpp -- the reference's accelerator seam (include/rakau/detail/rocm_fwd.hpp:22-46)
implemented on the rakau_amd C ABI -- compiled against the reference's own headers, with every instantiation of
src/rakau_rocm.cpp:333-370 (NDim {2,3} x F {float,double} x UInt {32,64 bit} x MAC {bh,bh_geom} x Q {0,1,2}).
Needs a checkout of the reference at /root/reference (only its HEADERS are read, at compile time; nothing of it enters
this repository or travels to the GPU box). The GPU test runs the driver built here, if it was built."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "integration"))
import build_bridges as bb  # noqa: E402  (the recipe, shared with __graft_entry__.build())

REF_INC, LIB, DRIVER, CUDA_LIB, CUDA_DRIVER = bb.REF_INC, bb.LIB, bb.DRIVER, bb.CUDA_LIB, bb.CUDA_DRIVER
build = bb.build


def driver_or_fail(path):
    """The GPU box has no checkout of the reference: the drivers travel prebuilt (tests/build/, made by
    __graft_entry__.build() where the reference's headers are). A missing driver there is a FAILURE, not a skip: these two
    tests are the only ones that call the engine through the reference-side bindings."""
    if bb.have_reference():
        bb.build(force=False)
    assert os.path.exists(path), ("%s was not built: run __graft_entry__.build() (or python integration/build_bridges.py) on a "
                                  "machine with the reference's headers at %s before shipping the tree to the GPU box" % (path, REF_INC))
    return path


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="no checkout of the reference: the bridge cannot be compiled")
def test_bridge_compiles_against_the_reference_headers():
    build()
    syms = subprocess.run(["nm", "-DC", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    for f in ("rakau::detail::rocm_min_size()", "rakau::detail::rocm_has_accelerator()", "rakau::detail::rakau_amd_set_ncrit("):
        assert f in syms
    for nd in (2, 3):
        for fp in ("float", "double"):
            for ui in ("unsigned int", "unsigned long"):
                for mac in ("0", "1"):
                    cls = r"rakau::detail::rocm_state<%dul, %s, %s, \(rakau::mac\)%s>::" % (nd, fp, ui, mac)
                    assert re.search(cls + r"rocm_state\(", syms) and re.search(cls + r"~rocm_state\(", syms), cls
                    for q in (0, 1, 2):
                        assert re.search(r"void " + cls + r"acc_pot<%du>\(" % q, syms), (cls, q)


@pytest.mark.gpu
def test_bridge_driver_on_gpu():
    out = subprocess.run([driver_or_fail(DRIVER)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bridge checks: 0 failure(s)" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="no checkout of the reference: the bridge cannot be compiled")
def test_cuda_bridge_compiles_against_the_reference_headers():
    """Every symbol of the CUDA seam the reference instantiates (src/rakau_cuda.cu:536-568: NDim {2,3} x F x UInt x Q x MAC =
    48 cuda_acc_pot_impl functions), cuda_min_size / cuda_device_count, and the two life-time hooks."""
    if not os.path.exists(CUDA_LIB):
        build()
    syms = subprocess.run(["nm", "-DC", "--defined-only", CUDA_LIB], capture_output=True, text=True, check=True).stdout
    for f in ("rakau::detail::cuda_min_size()", "rakau::detail::cuda_device_count()", "rakau::detail::rakau_amd_tree_ready(",
              "rakau::detail::rakau_amd_invalidate("):
        assert f in syms
    n = 0
    for nd in (2, 3):
        for fp in ("float", "double"):
            for ui in ("unsigned int", "unsigned long"):
                for mac in ("0", "1"):
                    for q in (0, 1, 2):
                        pat = r"void rakau::detail::cuda_acc_pot_impl<%du, %dul, %s, %s, \(rakau::mac\)%s>\(" % (q, nd, fp, ui, mac)
                        assert re.search(pat, syms), pat
                        n += 1
    assert n == 48


@pytest.mark.gpu
def test_cuda_bridge_driver_on_gpu():
    """cuda_acc_pot_impl over four logical devices (RK_ALIAS_DEVICES=4 on the 1-GPU box): every device share bit-identical
    to the one-device call, compact and offset outputs, announced (resident replicas, invalidated by a mass update) and
    unannounced (state per call) trees, 32-bit codes, the reference's error for too many accelerators."""
    env = dict(os.environ, RK_ALIAS_DEVICES="4")
    out = subprocess.run([driver_or_fail(CUDA_DRIVER)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "cuda bridge checks: 0 failure(s)" in out.stdout, out.stdout + out.stderr
