"""Python face of rakau_amd::octree<F, MAC> (include/rakau_amd/tree.hpp) via include/rakau_amd_tree.h.

Same names, argument meaning and error behaviour as the reference's ``rakau::octree`` acc/pot surface
(include/rakau/tree.hpp:3406-3497, 3572-3616, 3638-3837): ``accs_u/pots_u/accs_pots_u`` (+ ``_o``) with
keyword arguments ``G``, ``eps``, ``split``; ``exact_*``; ``perm/last_perm/inv_perm/nodes``;
``update_particles_u``. Construction and bookkeeping run on the host; every acc/pot call runs on the GPU.
"""
import ctypes as C

import numpy as np

from . import _capi
from .state import State, node_dtype, nres

_FP = {np.dtype(np.float32): _capi.RK_F32, np.dtype(np.float64): _capi.RK_F64}
_MAC = {"bh": _capi.RK_MAC_BH, "bh_geom": _capi.RK_MAC_BH_GEOM}


class Octree:
    """rakau::octree<F, MAC>; z_coords = None gives rakau::quadtree<F, MAC> (see Quadtree)."""

    def __init__(self, x_coords, y_coords, z_coords, masses, box_size=None, max_leaf_n=16, ncrit=128, mac="bh",
                 builder="host", code_bits=64):
        """code_bits = 32: tree<NDim, F, std::uint32_t, MAC> (10 / 15 bits per coordinate)."""
        if code_bits not in (32, 64):
            raise ValueError("code_bits must be 32 or 64")
        self.code_bits = code_bits
        coords = [x_coords, y_coords] + ([] if z_coords is None else [z_coords])
        arrs = [np.ascontiguousarray(v) for v in coords + [masses]]
        self.ndim = len(coords)
        x, m = arrs[0], arrs[-1]
        self.dtype = x.dtype
        if self.dtype not in _FP or any(v.dtype != self.dtype for v in arrs):
            raise TypeError("coordinates and masses must share a float32 or float64 dtype")
        if any(v.size != x.size for v in arrs[:-1]):
            raise ValueError("The input ranges for the particle coordinates have inconsistent sizes")
        if m.size != x.size:
            raise ValueError("The size of the input range for the particle masses (%d) is different from the size of "
                             "the input ranges for the particle coordinates (%d)" % (m.size, x.size))
        self.mac = mac
        self._h = C.c_void_p()
        if box_size is not None and box_size == 0:
            # An explicit zero box cannot hold particles; let the builder report it like the reference does.
            box_size = float(np.finfo(self.dtype).tiny)
        src = (C.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        _capi.check(_capi.lib().rk_tree_create_nd(C.byref(self._h), self.ndim, _FP[self.dtype], _MAC[mac], src, x.size,
                                                  0.0 if box_size is None else float(box_size), max_leaf_n, ncrit,
                                                  (1 if builder == "device" else 0) | (2 if code_bits == 32 else 0)))
        self._refresh()

    def _refresh(self):
        info = (C.c_int64 * 8)()
        box = C.c_double()
        _capi.check(_capi.lib().rk_tree_info(self._h, info, C.byref(box)))
        self.nparts, self.n_nodes, self.n_crit = int(info[0]), int(info[1]), int(info[2])
        self.max_leaf_n, self.ncrit = int(info[3]), int(info[4])
        self.box_size_deduced = bool(info[5])
        self.node_stride = int(info[6])
        self.box_size = box.value

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().rk_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- accessors -------------------------------------------------------------------------
    def _get(self, what, dtype, shape=None):
        out = np.empty(self.nparts if shape is None else shape, dtype=dtype)
        _capi.check(_capi.lib().rk_tree_get(self._h, what, out.ctypes.data))
        return out

    def p_its_u(self):
        """x, y, (z,) masses in Morton order."""
        return [self._get(k, self.dtype) for k in ((0, 1, 2, 3) if self.ndim == 3 else (0, 1, 3))]

    def c_it_u(self):
        return self._get(4, np.uint64 if self.code_bits == 64 else np.uint32)

    def perm(self):
        return self._get(5, np.uint64)

    def last_perm(self):
        return self._get(6, np.uint64)

    def inv_perm(self):
        return self._get(7, np.uint64)

    def crit_nodes(self):
        return self._get(8, np.uint64, (self.n_crit, 3))

    def nodes(self):
        """Copy of the node array with the reference's record layout (tree_fwd.hpp:77-116)."""
        ptr, cnt, stride = C.c_void_p(), C.c_int64(), C.c_int64()
        _capi.check(_capi.lib().rk_tree_nodes(self._h, C.byref(ptr), C.byref(cnt), C.byref(stride)))
        dt = node_dtype(self.dtype, self.mac, self.ndim, self.code_bits)
        assert dt.itemsize == stride.value
        if cnt.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (cnt.value * stride.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def state(self):
        """Device-resident state on GPU 0 (owned by the tree)."""
        h = C.c_void_p()
        _capi.check(_capi.lib().rk_tree_state(self._h, C.byref(h)))
        st = State._from_handle(h, self.dtype, self.mac)
        st._owned = False
        st._keepalive = self
        return st

    # ---- acc / pot -------------------------------------------------------------------------
    def _acc_pot(self, q, ordered, theta, G=1.0, eps=0.0, split=()):
        outs = [np.zeros(self.nparts, dtype=self.dtype) for _ in range(nres(q, self.ndim))]
        ptrs = (C.c_void_p * 4)(*[o.ctypes.data for o in outs], *([None] * (4 - len(outs))))
        sp = (C.c_double * max(1, len(split)))(*split)
        _capi.check(_capi.lib().rk_tree_acc_pot(self._h, q, int(ordered), ptrs, theta, G, eps, sp, len(split)))
        return outs

    def cpu_acc_pot_u(self, q, theta, G=1.0, eps=0.0, flavour="auto", nthreads=0):
        """tree::cpu_acc_pot_u: the header's CPU engine alone (no GPU involved). flavour: 'auto' | 'scalar' | 'simd_exact'."""
        outs = [np.zeros(self.nparts, dtype=self.dtype) for _ in range(nres(q, self.ndim))]
        ptrs = (C.c_void_p * 4)(*[o.ctypes.data for o in outs], *([None] * (4 - len(outs))))
        fl = {"auto": 0, "scalar": 1, "simd_exact": 2}[flavour]
        _capi.check(_capi.lib().rk_tree_cpu_acc_pot(self._h, q, ptrs, theta, G, eps, fl, nthreads))
        return outs

    def accs_u(self, theta, **kw):
        return self._acc_pot(0, False, theta, **kw)

    def pots_u(self, theta, **kw):
        return self._acc_pot(1, False, theta, **kw)[0]

    def accs_pots_u(self, theta, **kw):
        return self._acc_pot(2, False, theta, **kw)

    def accs_o(self, theta, **kw):
        return self._acc_pot(0, True, theta, **kw)

    def pots_o(self, theta, **kw):
        return self._acc_pot(1, True, theta, **kw)[0]

    def accs_pots_o(self, theta, **kw):
        return self._acc_pot(2, True, theta, **kw)

    def _exact(self, q, ordered, idx, G=1.0, eps=0.0):
        out = np.zeros(4, dtype=self.dtype)
        _capi.check(_capi.lib().rk_tree_exact(self._h, q, int(ordered), idx, G, eps, out.ctypes.data))
        return out[:nres(q, self.ndim)]

    def exact_acc_u(self, idx, **kw):
        return self._exact(0, False, idx, **kw)

    def exact_pot_u(self, idx, **kw):
        return self._exact(1, False, idx, **kw)[0]

    def exact_acc_pot_u(self, idx, **kw):
        return self._exact(2, False, idx, **kw)

    def exact_acc_o(self, idx, **kw):
        return self._exact(0, True, idx, **kw)

    def exact_pot_o(self, idx, **kw):
        return self._exact(1, True, idx, **kw)[0]

    def exact_acc_pot_o(self, idx, **kw):
        return self._exact(2, True, idx, **kw)

    # ---- updates ---------------------------------------------------------------------------
    def update_particles_u(self, func):
        """func receives [x, y, (z,) m] (numpy arrays in Morton order) and modifies them in place."""
        arrs = self.p_its_u()
        func(arrs)
        arrs = [np.ascontiguousarray(a, dtype=self.dtype) for a in arrs]
        ptrs = [a.ctypes.data for a in arrs]
        if self.ndim == 2:
            ptrs = [ptrs[0], ptrs[1], None, ptrs[2]]
        _capi.check(_capi.lib().rk_tree_update_particles(self._h, *ptrs))
        self._refresh()


class Quadtree(Octree):
    """rakau::quadtree<F, MAC>: the 2-dimensional variant (accelerations have two components)."""

    def __init__(self, x_coords, y_coords, masses, **kw):
        super().__init__(x_coords, y_coords, None, masses, **kw)
