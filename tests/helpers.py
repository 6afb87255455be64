"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
import numpy as np

import oracle
from rakau_amd import State, node_dtype, mac_value_of


def oracle_nodes_aos(ot):
    """Node array of an oracle tree in the reference's AoS layout (tree_fwd.hpp:77-116)."""
    nd = ot.nodes()
    dt = node_dtype(ot.dtype, ot.mac, getattr(ot, "ndim", 3))
    a = np.zeros(ot.n_nodes, dtype=dt)
    for k in ("begin", "end", "n_children", "code", "level"):
        a[k] = nd[k]
    a["props"] = nd["props"]
    if ot.mac == "bh":
        a["dim2"] = nd["dims"][:, 0]
    else:
        a["dim"] = nd["dims"][:, 0]
        a["delta"] = nd["dims"][:, 1]
    return a


def state_from_oracle(ot, device=0):
    p = ot.parts_u()
    if getattr(ot, "ndim", 3) == 2:
        return State(p[0], p[1], None, p[2], oracle_nodes_aos(ot), ncrit=ot.ncrit, mac=ot.mac, device=device)
    return State(p[0], p[1], p[2], p[3], oracle_nodes_aos(ot), ncrit=ot.ncrit, mac=ot.mac, device=device)


def rel_err_vec(a, b, ndim=3):
    """Per-particle |a - b| / |b| on ndim-vectors given as lists of arrays."""
    a = np.stack([np.asarray(v, dtype=np.float64) for v in a[:ndim]], axis=1)
    b = np.stack([np.asarray(v, dtype=np.float64) for v in b[:ndim]], axis=1)
    den = np.linalg.norm(b, axis=1)
    den[den == 0] = 1.0
    return np.linalg.norm(a - b, axis=1) / den


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b)
    den[den == 0] = 1.0
    return np.abs(a - b) / den
