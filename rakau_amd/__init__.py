"""rakau_amd: MI355X-native Barnes-Hut tree-traversal engine behind rakau's accs/pots API.

The package holds only what the hot path needs: ``csrc/`` (hand-written HIP kernels for gfx950 and the
C ABI of ``include/rakau_amd.h``), ``lib/`` (the built ``librakau_amd.so``) and thin ctypes plumbing.
"""
from . import _capi
from .state import State, node_dtype, mac_value_of, NRES, nres, pinned_empty
from .tree import Octree, Quadtree


def set_build_exact(on=True):
    """rk_set_build_exact: device-built trees with node properties bit-identical to the host builders' (slower build)."""
    _capi.lib().rk_set_build_exact(int(bool(on)))


__all__ = ["set_build_exact", "pinned_empty", "State", "Octree", "Quadtree", "node_dtype", "mac_value_of", "NRES", "nres"]
