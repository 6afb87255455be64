"""Device-resident traversal state: the Python face of rk_state_* / rk_acc_pot (include/rakau_amd.h).

Mirrors ``rakau::rocm_state<3, F, uint64_t, MAC>`` (include/rakau/detail/rocm_fwd.hpp:26-46 of the
reference): constructed from the Morton-ordered particle arrays and the node array of a built tree,
``acc_pot(q, p_begin, p_end, ...)`` fills caller-owned outputs.
"""
import ctypes as C

import numpy as np

from . import _capi

NRES = {0: 3, 1: 1, 2: 4}
_FP = {np.dtype(np.float32): _capi.RK_F32, np.dtype(np.float64): _capi.RK_F64}
_MAC = {"bh": _capi.RK_MAC_BH, "bh_geom": _capi.RK_MAC_BH_GEOM}


def nres(q, ndim=3):
    """Number of output arrays: ndim accelerations, 1 potential, or both (tree_nvecs_res)."""
    return {0: ndim, 1: 1, 2: ndim + 1}[q]


def node_dtype(fp_dtype, mac="bh", ndim=3, code_bits=64):
    """numpy structured dtype laid out as rakau::tree_node_t<ndim, F, UInt, MAC>
    (include/rakau/detail/tree_fwd.hpp:77-116 of the reference): 64/80 bytes (octree, bh), 64/88 (bh_geom) with 64-bit
    codes; code and level are UInt, the three sizes stay size_t."""
    f = np.dtype(fp_dtype)
    u = "<u8" if code_bits == 64 else "<u4"
    fields = [("begin", "<u8"), ("end", "<u8"), ("n_children", "<u8"), ("code", u), ("level", u),
              ("props", f, (ndim + 1,))]
    fields += [("dim2", f)] if mac == "bh" else [("dim", f), ("delta", f)]
    return np.dtype(fields, align=True)


def mac_value_of(theta, mac, dtype):
    """theta -> theta**-2 (bh) or theta**-1 (bh_geom) in the tree's precision (tree.hpp:3303-3312)."""
    t = np.dtype(dtype).type(theta)
    one = np.dtype(dtype).type(1)
    return float(one / (t * t)) if mac == "bh" else float(one / t)


class _PinnedBlock:
    """Owner of one rk_host_alloc() block (freed when the last numpy view of it dies)."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _capi.check(_capi.lib().rk_host_alloc(C.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes

    def __del__(self):
        try:
            if self.ptr:
                _capi.lib().rk_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(n, dtype):
    """A numpy array of n elements in pinned host memory (rk_host_alloc): as an output of State.acc_pot() / rk_acc_pot()
    it receives the results straight from the kernels (no staging copy)."""
    dtype = np.dtype(dtype)
    blk = _PinnedBlock(max(int(n), 1) * dtype.itemsize)
    buf = (C.c_char * blk.nbytes).from_address(blk.ptr)
    buf._rk_owner = blk  # the ctypes object is the array's base: keeps the block alive
    return np.frombuffer(buf, dtype=dtype, count=int(n))


class State:
    """z = None selects the 2-dimensional (quadtree) variant everywhere: coordinates are then x, y."""

    def __init__(self, x, y, z, m, nodes, ncrit=128, mac="bh", device=0, codes=None):
        lib = _capi.lib()
        arrs = [np.ascontiguousarray(v) for v in ((x, y, m) if z is None else (x, y, z, m))]
        ndim = len(arrs) - 1
        self.dtype = arrs[0].dtype
        if self.dtype not in _FP or any(v.dtype != self.dtype for v in arrs):
            raise TypeError("coordinates and masses must share a float32 or float64 dtype")
        if any(v.size != arrs[0].size for v in arrs):
            raise ValueError("The input ranges for the particle coordinates and masses have inconsistent sizes")
        if codes is not None:
            # rocm_state's ctor takes the sorted codes (rocm_fwd.hpp:29-30); this engine accepts and ignores them.
            codes = np.ascontiguousarray(codes, dtype=np.uint64)
            if codes.size != arrs[0].size:
                raise ValueError("codes must have one entry per particle")
        nodes = np.ascontiguousarray(nodes)
        if nodes.dtype != node_dtype(self.dtype, mac, ndim):
            raise TypeError("nodes must have dtype node_dtype(%s, %r, %d)" % (self.dtype, mac, ndim))
        self.mac = mac
        self._h = C.c_void_p()
        parts = (C.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        _capi.check(lib.rk_state_create_nd(C.byref(self._h), ndim, _FP[self.dtype], _MAC[mac], device, parts,
                                           codes.ctypes.data if codes is not None else None, arrs[0].size,
                                           nodes.ctypes.data, nodes.size, nodes.dtype.itemsize, ncrit))
        self._read_info()

    @classmethod
    def build(cls, x, y, z, m, box_size=None, max_leaf_n=16, ncrit=128, mac="bh", device=0):
        """Device-side tree construction (rk_state_build_nd): particles in the caller's original order."""
        arrs = [np.ascontiguousarray(v) for v in ((x, y, m) if z is None else (x, y, z, m))]
        dtype = arrs[0].dtype
        if dtype not in _FP or any(v.dtype != dtype for v in arrs):
            raise TypeError("coordinates and masses must share a float32 or float64 dtype")
        if any(v.size != arrs[0].size for v in arrs):
            raise ValueError("The input ranges for the particle coordinates have inconsistent sizes")
        h = C.c_void_p()
        parts = (C.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        _capi.check(_capi.lib().rk_state_build_nd(C.byref(h), len(arrs) - 1, _FP[dtype], _MAC[mac], device, parts, 0,
                                                  arrs[0].size, 0.0 if box_size is None else float(box_size),
                                                  max_leaf_n, ncrit))
        return cls._from_handle(h, dtype, mac)

    @classmethod
    def build_device(cls, d_ptrs, nparts, dtype, box_size=None, max_leaf_n=16, ncrit=128, mac="bh", device=0):
        """rk_state_build_nd(on_device): d_ptrs = DEVICE addresses (e.g. torch.Tensor.data_ptr()) of x, y, (z,) m --
        `nparts` values of `dtype` each, resident on `device`, in the caller's original order. Nothing crosses PCIe."""
        dtype = np.dtype(dtype)
        if dtype not in _FP:
            raise TypeError("dtype must be float32 or float64")
        h = C.c_void_p()
        parts = (C.c_void_p * 4)(*d_ptrs)
        _capi.check(_capi.lib().rk_state_build_nd(C.byref(h), len(d_ptrs) - 1, _FP[dtype], _MAC[mac], device, parts, 1,
                                                  nparts, 0.0 if box_size is None else float(box_size), max_leaf_n,
                                                  ncrit))
        return cls._from_handle(h, dtype, mac)

    def rebuild_device(self, d_ptrs, nparts=None, box_size=None):
        """rk_state_rebuild_device: new particle positions (device addresses, original order), same parameters."""
        nparts = self.nparts if nparts is None else nparts
        parts = (C.c_void_p * 4)(*d_ptrs)
        try:
            _capi.check(_capi.lib().rk_state_rebuild_device(self._h, parts, nparts,
                                                            0.0 if box_size is None else float(box_size)))
        finally:
            self._read_info()

    def set_perm(self, perm):
        """Install tree::perm() (uint64, host) so that ordered=True outputs work on a state made from a host tree."""
        perm = np.ascontiguousarray(perm, dtype=np.uint64)
        if perm.size != self.nparts:
            raise ValueError("perm must have one entry per particle")
        _capi.check(_capi.lib().rk_state_set_perm(self._h, perm.ctypes.data))

    def device_ptr(self, what):
        """(address, bytes) of a resident array: 'parts' ({x,y,z,m} AoS, Morton order), 'perm' (uint32), 'codes',
        'first_order' (launch order of the first call: critical-node indices, uint32), 'first_tab' (queue table of that order,
        72 uint32: eight queues for a small tree, queues per class and region for a large one; (0, 0) unless the tree came with one)."""
        ptr = C.c_void_p()
        nbytes = C.c_int64()
        sel = {"parts": 0, "perm": 1, "codes": 2, "first_order": 3, "first_tab": 4}[what]
        _capi.check(_capi.lib().rk_state_device_ptr(self._h, sel, C.byref(ptr), C.byref(nbytes)))
        return ptr.value or 0, nbytes.value

    def tree_info(self):
        box = C.c_double()
        info = (C.c_int64 * 4)()
        _capi.check(_capi.lib().rk_state_tree_info(self._h, C.byref(box), info))
        return dict(box_size=box.value, box_deduced=bool(info[0]), max_leaf_n=int(info[1]), device_built=bool(info[2]),
                    n_internal=int(info[3]))

    def download(self, what):
        """what: 'x','y','z','m' (Morton order), 'codes', 'perm', 'nodes' (reference record layout), 'crit'."""
        sel = {"x": 0, "y": 1, "z": 2, "m": 3, "codes": 4, "perm": 5, "nodes": 6, "crit": 7}[what]
        if sel <= 3:
            out = np.empty(self.nparts, dtype=self.dtype)
        elif sel in (4, 5):
            out = np.empty(self.nparts, dtype=np.uint64)
        elif sel == 6:
            out = np.zeros(self.tree_size, dtype=node_dtype(self.dtype, self.mac, self.ndim))
        else:
            out = np.empty((self.n_crit, 3), dtype=np.uint64)
        _capi.check(_capi.lib().rk_state_download(self._h, sel, out.ctypes.data))
        return out

    @classmethod
    def _from_handle(cls, handle, dtype, mac):
        self = cls.__new__(cls)
        self._h = handle
        self.dtype = np.dtype(dtype)
        self.mac = mac
        self._owned = True
        self._read_info()
        return self

    def _read_info(self):
        info = (C.c_int64 * 8)()
        _capi.check(_capi.lib().rk_state_info(self._h, info))
        self.nparts, self.tree_size, self.n_crit, self.max_group = (int(v) for v in info[:4])
        self.device = int(info[6])
        self.ncrit = int(info[7])
        self.ndim = int(_capi.lib().rk_state_ndim(self._h))

    def close(self):
        if getattr(self, "_h", None) and getattr(self, "_owned", True):
            _capi.lib().rk_state_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def crit_ranges(self):
        out = np.empty((self.n_crit, 2), dtype=np.int64)
        _capi.check(_capi.lib().rk_state_crit_ranges(self._h, out.ctypes.data))
        return out

    def set_variant(self, v):
        _capi.check(_capi.lib().rk_set_kernel_variant(self._h, v))

    def graph_stats(self):
        """rk_state_graph_stats: dict(replays, captures, direct, retargeted, cached, forked_alive)."""
        a = (C.c_int64 * 6)()
        _capi.check(_capi.lib().rk_state_graph_stats(self._h, a))
        return dict(zip(("replays", "captures", "direct", "retargeted", "cached", "forked_alive"), (int(v) for v in a)))

    def acc_pot(self, q, mac_value, G=1.0, eps2=0.0, p_begin=0, p_end=None, out=None, offset_output=True, ordered=False):
        """rocm_state::acc_pot<Q>: host outputs (numpy). Returns the list of output arrays. Arrays in pinned memory
        (pinned_empty()) are written by the kernels directly. ordered=True (whole range only, after set_perm()): the
        result of every particle at its ORIGINAL index (accs_o / pots_o), scattered on the device."""
        p_end = self.nparts if p_end is None else p_end
        if out is None:
            n = self.nparts if (offset_output or ordered) else p_end - p_begin
            out = [np.zeros(n, dtype=self.dtype) for _ in range(nres(q, self.ndim))]
        ptrs = (C.c_void_p * 4)(*[o.ctypes.data for o in out], *([None] * (4 - len(out))))
        flags = (_capi.RK_OUT_OFFSET | _capi.RK_OUT_ORDERED) if ordered else int(bool(offset_output))
        _capi.check(_capi.lib().rk_acc_pot(self._h, q, p_begin, p_end, ptrs, mac_value, G, eps2, flags))
        return out

    def acc_pot_device(self, q, mac_value, d_ptrs, G=1.0, eps2=0.0, p_begin=0, p_end=None, offset_output=True,
                       stream=None, ordered=False):
        """Outputs stay in HBM: d_ptrs are device addresses (e.g. torch.Tensor.data_ptr()). ordered=True writes the
        result of every particle at its ORIGINAL index (accs_o / pots_o semantics; full-size outputs)."""
        p_end = self.nparts if p_end is None else p_end
        ptrs = (C.c_void_p * 4)(*d_ptrs, *([None] * (4 - len(d_ptrs))))
        flags = _capi.RK_OUT_ORDERED if ordered else int(bool(offset_output))
        _capi.check(_capi.lib().rk_acc_pot_device(self._h, q, p_begin, p_end, ptrs, mac_value, G, eps2, flags,
                                                  stream))

    def count_interactions(self, mac_value, p_begin=0, p_end=None):
        """Census of the traversal: dict(mac=..., com=..., pp=..., self=...) particle-level counts."""
        p_end = self.nparts if p_end is None else p_end
        c = (C.c_uint64 * 4)()
        _capi.check(_capi.lib().rk_count_interactions(self._h, p_begin, p_end, mac_value, c))
        return dict(mac=int(c[0]), com=int(c[1]), pp=int(c[2]), self=int(c[3]))

    def group_work(self, mac_value):
        """Particle-level interactions per critical node (uint64[n_crit]): the load-balancing weight of each group."""
        out = np.zeros(self.n_crit, dtype=np.uint64)
        _capi.check(_capi.lib().rk_group_work(self._h, mac_value, out.ctypes.data))
        return out

    def set_timing(self, on):
        """rk_state_set_timing: record (or not) the HIP events behind last_kernel_ms() around every device-output call."""
        _capi.check(_capi.lib().rk_state_set_timing(self._h, int(bool(on))))

    def last_kernel_ms(self):
        ms = C.c_float()
        _capi.check(_capi.lib().rk_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def export(self):
        """(ptrs, bytes, meta) of the device buffers that make up the state (for replication)."""
        cnt = C.c_int()
        ptrs = (C.c_void_p * _capi.RK_MAX_BUFFERS)()
        nbytes = (C.c_int64 * _capi.RK_MAX_BUFFERS)()
        meta = (C.c_int64 * _capi.RK_META_WORDS)()
        _capi.check(_capi.lib().rk_state_export(self._h, C.byref(cnt), ptrs, nbytes, meta))
        n = cnt.value
        return [ptrs[i] or 0 for i in range(n)], [int(nbytes[i]) for i in range(n)], [int(v) for v in meta]

    def clone(self, device):
        """rk_state_clone: replica on another device of this process (device-to-device / xGMI peer copies)."""
        h = C.c_void_p()
        _capi.check(_capi.lib().rk_state_clone(C.byref(h), self._h, device))
        return State._from_handle(h, self.dtype, self.mac)

    def clone_all(self, devices):
        """rk_state_clone_all: replicas on several devices of this process at once (doubling tree of peer copies)."""
        n = len(devices)
        outs = (C.c_void_p * n)()
        devs = (C.c_int * n)(*devices)
        _capi.check(_capi.lib().rk_state_clone_all(outs, self._h, devs, n))
        return [State._from_handle(C.c_void_p(outs[i]), self.dtype, self.mac) for i in range(n)]

    @classmethod
    def broadcast(cls, state, root, rank, device, comm, dtype=None, mac=None, stream=None):
        """rk_state_broadcast: `state` of rank `root` is replicated on every rank of the RCCL communicator `comm` (Comm below).
        Returns the state of this rank (the root's own on the root)."""
        h = C.c_void_p(state._h.value if state is not None else None)
        _capi.check(_capi.lib().rk_state_broadcast(C.byref(h), root, rank, device, comm.handle, stream))
        if rank == root:
            return state
        info = (C.c_int64 * 8)()
        _capi.check(_capi.lib().rk_state_info(h, info))
        fp = int(info[4])
        mac_id = int(info[5])
        return cls._from_handle(h, np.float32 if fp == _capi.RK_F32 else np.float64, "bh" if mac_id == _capi.RK_MAC_BH else "bh_geom")

    @classmethod
    def from_buffers(cls, device, ptrs, nbytes, meta):
        h = C.c_void_p()
        n = len(ptrs)
        cp = (C.c_void_p * n)(*ptrs)
        cb = (C.c_int64 * n)(*nbytes)
        cm = (C.c_int64 * _capi.RK_META_WORDS)(*meta)
        _capi.check(_capi.lib().rk_state_import(C.byref(h), device, n, cp, cb, cm))
        dtype = np.float32 if meta[1] == _capi.RK_F32 else np.float64
        mac = "bh" if meta[2] == _capi.RK_MAC_BH else "bh_geom"
        return cls._from_handle(h, dtype, mac)


class Comm:
    """RCCL communicator made by the library (rk_comm_*): Comm.unique_id() on one rank, ship the bytes, Comm(n, id, rank, device)
    on every rank."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _capi.check(_capi.lib().rk_comm_unique_id(buf))
        return buf.raw

    def __init__(self, n_ranks, uid, rank, device):
        self.handle = C.c_void_p()
        _capi.check(_capi.lib().rk_comm_init(C.byref(self.handle), n_ranks, uid, rank, device))

    def close(self):
        if self.handle:
            _capi.lib().rk_comm_destroy(self.handle)
            self.handle = C.c_void_p()
