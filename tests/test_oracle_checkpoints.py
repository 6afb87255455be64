"""Pin the CPU oracle against the reference outputs recorded in SURVEY.md section 8(c)
(tests/golden/survey_checkpoints.json): node counts, deduced box size, accs_u at Morton index 0 and the
interaction census for the default-seeded benchmark Plummer sphere."""
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "survey_checkpoints.json")) as f:
    CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d" % c["nparts"])
def test_survey_checkpoint(case):
    n = case["nparts"]
    m, x, y, z = oracle.plummer(n, np.float32)  # default-seeded mt19937, benchmark/common.hpp:36
    t = oracle.Tree(x, y, z, m)
    assert t.n_nodes == case["n_nodes"]
    assert t.n_crit == case["n_crit"]
    assert abs(t.box_size - case["box_size"]) <= 6e-6 * case["box_size"]  # recorded with 6 digits
    # Only the first critical node is traversed: index 0 lives there.
    full = n <= 100000
    outs, st = t.acc_pot(0, 0.75, nthreads=8, c_begin=0, c_end=(2 ** 62 if full else 1), want_stats=True)
    got = np.array([o[0] for o in outs], dtype=np.float64)
    ref = np.array(case["accs_u_index0"])
    # The checkpoints were printed with 9 significant digits from an -O3 -march=native build (whose FMA
    # contraction choices are the compiler's); agreement to a few fp32 ulps is the pin.
    assert np.all(np.abs(got - ref) <= 2e-6 * np.abs(ref)), (got, ref)
    if full:
        per = case["interactions_per_particle"]
        assert round(st["w_com"] / n) == per["com"]
        assert round(st["w_pp"] / n) == per["pp"]
        assert round((st["w_com"] + st["w_pp"] + st["w_self"]) / n) == per["total"]
