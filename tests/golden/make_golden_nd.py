#!/usr/bin/env python3
"""Generate tests/golden/oracle_cases_nd.npz: frozen vectors for the quadtree (NDim = 2) and 32-bit-code variants.

Same provenance as make_golden.py (PARITY-UNPINNED): produced by the CPU oracle, not by the reference (which cannot be
built here); for quadtrees the oracle is constrained -- not pinned -- by the reference's node-centre known-answer test and
by the accuracy / G / ordering properties restated on it (tests/test_oracle_quadtree.py).

Run from the repository root:  python tests/golden/make_golden_nd.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

CASES = [
    # name, ndim, code_bits, n, dtype, mac, theta, eps, G, max_leaf_n, ncrit
    ("quad_f32_bh", 2, 64, 2500, np.float32, "bh", 0.75, 0.0, 1.0, 16, 128),
    ("quad_f64_geom", 2, 64, 2500, np.float64, "bh_geom", 0.5, 0.02, 2.0, 4, 32),
    ("oct_u32_f64_bh", 3, 32, 3000, np.float64, "bh", 0.6, 0.01, 1.0, 2, 64),
    ("quad_u32_f32_geom", 2, 32, 3000, np.float32, "bh_geom", 0.75, 0.05, 0.5, 8, 128),
]


def main():
    out = {}
    for name, ndim, bits, n, dtype, mac, theta, eps, G, mln, ncrit in CASES:
        p = oracle.Rng(77).uniform_particles(n, 1.0, dtype, ndim=ndim)
        m, c = p[0], list(p[1:])
        t = oracle.Tree(c[0], c[1], c[2] if ndim == 3 else None, m, max_leaf_n=mln, ncrit=ncrit, mac=mac, ndim=ndim,
                        code_bits=bits)
        res = t.accs_pots_o(theta, eps=eps, G=G)
        out[name + "/in"] = np.stack(c + [m])
        out[name + "/out"] = np.stack(res)
        out[name + "/meta"] = np.array([ndim, bits, theta, eps, G, mln, ncrit, t.n_nodes, t.n_crit, t.box_size])
        out[name + "/mac"] = np.array(mac)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_cases_nd.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
