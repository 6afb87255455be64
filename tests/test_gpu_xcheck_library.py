"""The cross-check kernels (variant 1: scalar depth-first walk, variant 4: split traversal) live in librakau_amd_xcheck.so, which
librakau_amd.so loads from its own directory the first time one of them is selected. Without that file the selection fails
loudly -- RK_ERUNTIME naming the path -- and the default path is untouched; with it, the variants agree with the default kernel."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = """
import sys, numpy as np, oracle, rakau_amd
from helpers import state_from_oracle, rel_err_vec
m, x, y, z = oracle.plummer(20000, np.float32)
ot = oracle.Tree(x, y, z, m)
st = state_from_oracle(ot)
mv = rakau_amd.mac_value_of(0.75, "bh", np.float32)
base = st.acc_pot(0, mv)
for v in (1, 4):
    try:
        st.set_variant(v)
    except RuntimeError as e:
        print("VARIANT %d REFUSED: %s" % (v, e))
        continue
    got = st.acc_pot(0, mv)
    print("VARIANT %d OK max rel diff %.3g" % (v, rel_err_vec(got, base).max()))
st.set_variant(0)
again = st.acc_pot(0, mv)
print("DEFAULT SAME BITS", all(np.array_equal(a, b) for a, b in zip(base, again)))
"""


def run(libdir):
    env = dict(os.environ, RAKAU_AMD_LIB=os.path.join(libdir, "librakau_amd.so"))
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def test_cross_check_variants_need_their_library(tmp_path):
    src = os.path.join(ROOT, "rakau_amd", "lib")
    # A deployment without the cross-check library: the product library (and its AVX-512 CPU engine) only.
    bare = tmp_path / "bare"
    bare.mkdir()
    for f in ("librakau_amd.so", "librakau_amd_cpu512.so"):
        shutil.copy(os.path.join(src, f), bare / f)
    out = run(str(bare))
    assert "VARIANT 1 REFUSED" in out and "VARIANT 4 REFUSED" in out and "librakau_amd_xcheck.so" in out
    assert "DEFAULT SAME BITS True" in out
    # The full set: both variants run and agree with the default kernel to rounding (their summation orders differ).
    full = tmp_path / "full"
    full.mkdir()
    for f in ("librakau_amd.so", "librakau_amd_cpu512.so", "librakau_amd_xcheck.so"):
        shutil.copy(os.path.join(src, f), full / f)
    out = run(str(full))
    for v in (1, 4):
        line = [l for l in out.splitlines() if l.startswith("VARIANT %d OK" % v)]
        assert line, out
        assert float(line[0].split()[-1]) < 2e-5
    assert "DEFAULT SAME BITS True" in out
