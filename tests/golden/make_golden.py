#!/usr/bin/env python3
"""Generate tests/golden/oracle_cases.npz: small input/output vectors for the hot path.

PROVENANCE: PARITY-UNPINNED, as oracle/rakau_oracle.cpp:14 says. The reference cannot be built or run in this image
(its mandatory dependencies TBB, xsimd and Boost are absent) and ships no input/output vectors, so these vectors are
produced by the CPU oracle (oracle/rakau_oracle.cpp), a line-level restatement of the reference's scalar CPU path. The
oracle is constrained by the reference's own property tests restated on it (tests/test_oracle_reference_tests.py); the
numbers in tests/golden/survey_checkpoints.json come from a survey-time build against stand-in headers and pin nothing.
These files freeze the oracle's behaviour so that drift is detected, and give the GPU tests fixed vectors that do not
depend on the oracle library being rebuilt. They are not outputs of the reference.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

CASES = [
    # name, generator, n, dtype, mac, theta, eps, G, max_leaf_n, ncrit
    ("plummer_f32_bh", "plummer", 3000, np.float32, "bh", 0.75, 0.0, 1.0, 16, 128),
    ("plummer_f64_bh", "plummer", 3000, np.float64, "bh", 0.5, 0.01, 1.0, 16, 128),
    ("uniform_f32_geom", "uniform", 2000, np.float32, "bh_geom", 0.75, 0.05, 2.0, 8, 16),
    ("uniform_f64_geom", "uniform", 2000, np.float64, "bh_geom", 0.4, 0.0, 0.5, 2, 256),
]


def main():
    out = {}
    for name, gen, n, dtype, mac, theta, eps, G, mln, ncrit in CASES:
        if gen == "plummer":
            m, x, y, z = oracle.plummer(n, dtype)
        else:
            m, x, y, z = oracle.Rng(42).uniform_particles(n, 1.0, dtype)
        t = oracle.Tree(x, y, z, m, max_leaf_n=mln, ncrit=ncrit, mac=mac)
        res = t.accs_pots_o(theta, eps=eps, G=G)
        out[name + "/in"] = np.stack([x, y, z, m])
        out[name + "/out"] = np.stack(res)
        out[name + "/meta"] = np.array([theta, eps, G, mln, ncrit, t.n_nodes, t.n_crit, t.box_size])
        out[name + "/mac"] = np.array(mac)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_cases.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
