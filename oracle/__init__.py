"""ORACLE -- test infrastructure only (see oracle/rakau_oracle.cpp header).

ctypes binding of ``liboracle.so``, the CPU restatement of rakau's Barnes-Hut path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product package ``rakau_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_FP = {np.dtype(np.float32): 0, np.dtype(np.float64): 1}
_NP = {0: np.float32, 1: np.float64}
MAC = {"bh": 0, "bh_geom": 1}
NRES = {0: 3, 1: 1, 2: 4}


def nres(q, ndim=3):
    """Number of output arrays: ndim accelerations, 1 potential, or both (tree_fwd.hpp: tree_nvecs_res)."""
    return {0: ndim, 1: 1, 2: ndim + 1}[q]


class OracleError(Exception):
    pass


_EXC = {1: ValueError, 2: ArithmeticError, 3: OverflowError, 4: RuntimeError}


def build(force=False):
    """Compile liboracle.so with the committed recipe (oracle/Makefile)."""
    src = os.path.join(_HERE, "rakau_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_plummer.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint]
        L.orc_rng_create.restype = C.c_void_p
        L.orc_rng_create.argtypes = [C.c_uint]
        L.orc_rng_destroy.argtypes = [C.c_void_p]
        L.orc_uniform.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
        L.orc_uniform_nd.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
        L.orc_tree_create_ex.restype = C.c_void_p
        L.orc_tree_create_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_uint64,
                                         C.c_double, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_tree_create_nd.restype = C.c_void_p
        L.orc_tree_create_nd.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_uint64, C.c_double,
                                         C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_tree_create.restype = C.c_void_p
        L.orc_tree_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                      C.c_double, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_tree_destroy.argtypes = [C.c_void_p]
        L.orc_set_simd_width.argtypes = [C.c_int]
        L.orc_tree_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.orc_tree_get_parts.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.orc_tree_get_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tree_get_crit.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_acc_pot.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_uint, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_exact.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_double, C.c_double]
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise _EXC.get(rc, RuntimeError)(lib().orc_last_error().decode())


def plummer(n, dtype=np.float32, a=1.0, size=0.0, seed=5489):
    """benchmark/common.hpp:39-126 (serial branch). Returns (m, x, y, z). seed 5489 = default-seeded mt19937."""
    out = np.empty(4 * n, dtype=dtype)
    _check(lib().orc_plummer(_FP[np.dtype(dtype)], out.ctypes.data, n, a, size, seed))
    return out[:n], out[n:2 * n], out[2 * n:3 * n], out[3 * n:]


class Rng:
    """A std::mt19937 whose stream continues across calls (test/accuracy_acc.cpp:39)."""

    def __init__(self, seed):
        self._h = lib().orc_rng_create(seed)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_rng_destroy(self._h)
            self._h = None

    def uniform_particles(self, n, size, dtype, ndim=3):
        """test/test_utils.hpp:41-59 (get_uniform_particles<ndim>). Returns (m, x, y, z) or (m, x, y)."""
        out = np.empty((ndim + 1) * n, dtype=dtype)
        _check(lib().orc_uniform_nd(ndim, _FP[np.dtype(dtype)], out.ctypes.data, n, float(size), self._h))
        return tuple(out[k * n:(k + 1) * n] for k in range(ndim + 1))


def set_simd_width(w):
    """Association of the node sums of octrees built from now on: 1 = the reference's scalar build (default), 4 / 8 / 16 = its
    SIMD build with that batch size (tree.hpp:1134-1161: interleaved partial sums + horizontal add + scalar tail)."""
    lib().orc_set_simd_width(int(w))


class Tree:
    """CPU restatement of rakau::octree<F, MAC> / quadtree<F, MAC> (construction + acc/pot + exact sums).
    ndim = 2: pass z = None; everything that lists coordinates then has two of them."""

    def __init__(self, x, y, z, m, box_size=0.0, max_leaf_n=16, ncrit=128, mac="bh", ndim=3, code_bits=64):
        """code_bits = 32 restates tree<NDim, F, std::uint32_t, MAC>: 10 (3-D) / 15 (2-D) bits per coordinate."""
        assert ndim in (2, 3) and (z is None) == (ndim == 2) and code_bits in (32, 64)
        self.code_bits = code_bits
        arrs = [np.ascontiguousarray(v) for v in ((x, y, m) if ndim == 2 else (x, y, z, m))]
        self.ndim = ndim
        self.dtype = arrs[0].dtype
        assert all(v.dtype == self.dtype and v.size == arrs[0].size for v in arrs)
        self.fp = _FP[self.dtype]
        self.mac = mac
        st = C.c_int(0)
        src = (C.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        self._h = lib().orc_tree_create_ex(ndim, code_bits, self.fp, MAC[mac], src, arrs[0].size, float(box_size),
                                           max_leaf_n, ncrit, C.byref(st))
        _check(st.value)
        info = (C.c_uint64 * 4)()
        box = C.c_double()
        lib().orc_tree_info(self._h, info, C.byref(box))
        self.nparts, self.n_nodes, self.n_crit = int(info[0]), int(info[1]), int(info[2])
        self.box_size = box.value
        self.max_leaf_n, self.ncrit = max_leaf_n, ncrit

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().orc_tree_destroy(self._h)
            except TypeError:  # interpreter shutdown: the module globals are gone already
                pass
            self._h = None

    def parts_u(self):
        n = self.nparts
        out = [np.empty(n, dtype=self.dtype) for _ in range(4)]
        lib().orc_tree_get_parts(self._h, *[o.ctypes.data for o in out], None, None, None, None)
        return out if self.ndim == 3 else [out[0], out[1], out[3]]  # x, y, (z,) m in Morton order

    def codes_perms(self):
        n = self.nparts
        out = [np.empty(n, dtype=np.uint64) for _ in range(4)]
        lib().orc_tree_get_parts(self._h, None, None, None, None, *[o.ctypes.data for o in out])
        return dict(codes=out[0], perm=out[1], last_perm=out[2], inv_perm=out[3])

    def nodes(self):
        n = self.n_nodes
        topo = np.empty((n, 5), dtype=np.uint64)
        props = np.empty((n, self.ndim + 1), dtype=self.dtype)
        dims = np.empty((n, 2), dtype=self.dtype)
        lib().orc_tree_get_nodes(self._h, topo.ctypes.data, props.ctypes.data, dims.ctypes.data)
        return dict(begin=topo[:, 0].copy(), end=topo[:, 1].copy(), n_children=topo[:, 2].copy(),
                    code=topo[:, 3].copy(), level=topo[:, 4].copy(), props=props, dims=dims)

    def crit_nodes(self):
        c = np.empty((self.n_crit, 3), dtype=np.uint64)
        lib().orc_tree_get_crit(self._h, c.ctypes.data)
        return c  # code, begin, end

    def acc_pot(self, q, theta, G=1.0, eps=0.0, ordered=False, nthreads=1, c_begin=0, c_end=2 ** 62,
                want_stats=False):
        n = self.nparts
        outs = [np.zeros(n, dtype=self.dtype) for _ in range(nres(q, self.ndim))]
        ptrs = (C.c_void_p * 4)(*[o.ctypes.data for o in outs], *([None] * (4 - len(outs))))
        stats = np.zeros(9, dtype=np.uint64)
        _check(lib().orc_acc_pot(self._h, q, ptrs, int(ordered), theta, G, eps, nthreads, c_begin, c_end,
                                 stats.ctypes.data if want_stats else None))
        if want_stats:
            return outs, dict(zip(("visits", "com", "leaves", "pp", "self_pairs", "w_visits", "w_com", "w_pp", "w_self"),
                                   (int(s) for s in stats)))
        return outs

    def accs_u(self, theta, **kw):
        return self.acc_pot(0, theta, ordered=False, **kw)

    def pots_u(self, theta, **kw):
        return self.acc_pot(1, theta, ordered=False, **kw)[0]

    def accs_pots_u(self, theta, **kw):
        return self.acc_pot(2, theta, ordered=False, **kw)

    def accs_o(self, theta, **kw):
        return self.acc_pot(0, theta, ordered=True, **kw)

    def pots_o(self, theta, **kw):
        return self.acc_pot(1, theta, ordered=True, **kw)[0]

    def accs_pots_o(self, theta, **kw):
        return self.acc_pot(2, theta, ordered=True, **kw)

    def exact(self, q, idx, G=1.0, eps=0.0, ordered=False):
        out = np.zeros(4, dtype=self.dtype)
        _check(lib().orc_exact(self._h, q, out.ctypes.data, int(ordered), idx, G, eps))
        return out[:nres(q, self.ndim)]
