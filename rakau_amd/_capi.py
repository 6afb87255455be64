"""ctypes binding of librakau_amd.so (the C ABI declared in include/rakau_amd.h).

Plumbing only: loads the in-tree shared library and declares the prototypes. The product path fails
loudly if the HIP library is missing -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAKAU_AMD_LIB selects an alternative build of the same library (diagnostic builds).
LIB_PATH = os.environ.get("RAKAU_AMD_LIB") or os.path.join(_HERE, "lib", "librakau_amd.so")

RK_F32, RK_F64 = 0, 1
RK_MAC_BH, RK_MAC_BH_GEOM = 0, 1
RK_OUT_COMPACT, RK_OUT_OFFSET, RK_OUT_ORDERED = 0, 1, 2
RK_MAX_BUFFERS, RK_META_WORDS = 16, 32

# Status code -> Python exception mirroring the C++ exception types of the reference
# (std::invalid_argument, std::domain_error, std::overflow_error, std::runtime_error, std::bad_alloc).
_EXC = {1: ValueError, 2: ArithmeticError, 3: OverflowError, 4: RuntimeError, 5: MemoryError}

# Every symbol include/rakau_amd.h declares (checked by tests/test_capi_symbols.py).
SYMBOLS = [
    "rk_last_error", "rk_min_size", "rk_has_accelerator", "rk_device_count", "rk_state_create", "rk_state_destroy",
    "rk_state_info", "rk_state_crit_ranges", "rk_acc_pot", "rk_acc_pot_device", "rk_last_kernel_ms", "rk_state_export",
    "rk_state_import", "rk_state_clone", "rk_set_kernel_variant", "rk_device_memcpy", "rk_count_interactions", "rk_state_build",
    "rk_state_tree_info", "rk_state_download", "rk_state_build_device", "rk_state_set_perm", "rk_state_device_ptr",
    "rk_state_rebuild_device", "rk_pool_trim", "rk_set_build_exact", "rk_cpu_engine_run", "rk_group_work", "rk_state_create_nd", "rk_state_build_nd",
    "rk_state_ndim", "rk_host_alloc", "rk_host_free", "rk_state_set_timing", "rk_state_clone_all", "rk_comm_unique_id",
    "rk_comm_init", "rk_comm_destroy", "rk_state_broadcast", "rk_init", "rk_state_graph_stats",
    # host-side tree builder (include/rakau_amd_tree.h)
    "rk_tree_create", "rk_tree_create_nd", "rk_tree_destroy", "rk_tree_info", "rk_tree_get", "rk_tree_nodes", "rk_tree_state",
    "rk_tree_acc_pot", "rk_tree_exact", "rk_tree_update_particles", "rk_tree_cpu_acc_pot",
]

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same
    sonames as /opt/rocm's, requested by torch under the unversioned file names). If librakau_amd.so pulls in the system
    runtime first, a later `import torch` maps a SECOND runtime and its device initialisation fails ("No HIP GPUs are
    available"); loaded in the other order the dynamic linker resolves our libamdhip64.so.7 to torch's copy and both
    share it. So when a torch with bundled runtime is installed (and not yet imported), map its runtime first -- without
    importing torch. RAKAU_AMD_SYSTEM_HIP=1 opts out."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("RAKAU_AMD_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "rakau_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C rakau_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, i64, u64, dbl, ci = C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int
    L.rk_last_error.restype = C.c_char_p
    L.rk_min_size.restype = C.c_uint
    L.rk_state_create.argtypes = [C.POINTER(vp), ci, ci, ci, C.POINTER(vp), vp, i64, vp, i64, i64, u64]
    L.rk_state_destroy.argtypes = [vp]
    L.rk_state_destroy.restype = None
    L.rk_state_info.argtypes = [vp, C.POINTER(i64)]
    L.rk_state_crit_ranges.argtypes = [vp, vp]
    L.rk_acc_pot.argtypes = [vp, ci, i64, i64, C.POINTER(vp), dbl, dbl, dbl, ci]
    L.rk_acc_pot_device.argtypes = [vp, ci, i64, i64, C.POINTER(vp), dbl, dbl, dbl, ci, vp]
    L.rk_host_alloc.argtypes = [C.POINTER(vp), i64]
    L.rk_host_free.argtypes = [vp]
    L.rk_state_set_timing.argtypes = [vp, ci]
    L.rk_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.rk_state_export.argtypes = [vp, C.POINTER(ci), C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    L.rk_state_import.argtypes = [C.POINTER(vp), ci, ci, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    L.rk_state_clone.argtypes = [C.POINTER(vp), vp, ci]
    L.rk_init.argtypes = [ci]
    L.rk_state_clone_all.argtypes = [C.POINTER(vp), vp, C.POINTER(ci), ci]
    L.rk_comm_unique_id.argtypes = [C.c_char_p]
    L.rk_comm_init.argtypes = [C.POINTER(vp), ci, C.c_char_p, ci, ci]
    L.rk_comm_destroy.argtypes = [vp]
    L.rk_state_broadcast.argtypes = [C.POINTER(vp), ci, ci, ci, vp, vp]
    L.rk_set_kernel_variant.argtypes = [vp, ci]
    if hasattr(L, "rk_state_graph_stats"):  # (absent from older diagnostic builds selected with RAKAU_AMD_LIB)
        L.rk_state_graph_stats.argtypes = [vp, C.POINTER(i64)]
    L.rk_device_memcpy.argtypes = [vp, vp, i64, ci]
    L.rk_count_interactions.argtypes = [vp, i64, i64, dbl, C.POINTER(u64)]
    L.rk_state_build.argtypes = [C.POINTER(vp), ci, ci, ci, C.POINTER(vp), i64, dbl, u64, u64]
    L.rk_state_tree_info.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    L.rk_state_download.argtypes = [vp, ci, vp]
    L.rk_state_build_device.argtypes = [C.POINTER(vp), ci, ci, ci, C.POINTER(vp), i64, dbl, u64, u64]
    L.rk_state_set_perm.argtypes = [vp, vp]
    L.rk_state_rebuild_device.argtypes = [vp, C.POINTER(vp), i64, dbl]
    L.rk_group_work.argtypes = [vp, dbl, vp]
    L.rk_state_create_nd.argtypes = [C.POINTER(vp), ci, ci, ci, ci, C.POINTER(vp), vp, i64, vp, i64, i64, u64]
    L.rk_state_build_nd.argtypes = [C.POINTER(vp), ci, ci, ci, ci, C.POINTER(vp), ci, i64, dbl, u64, u64]
    L.rk_state_ndim.argtypes = [vp]
    L.rk_cpu_engine_run.argtypes = [vp]
    L.rk_set_build_exact.argtypes = [ci]
    L.rk_set_build_exact.restype = None
    L.rk_pool_trim.argtypes = []
    L.rk_pool_trim.restype = None
    L.rk_state_device_ptr.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(i64)]
    if hasattr(L, "rk_tree_create"):
        L.rk_tree_create.argtypes = [C.POINTER(vp), ci, ci, vp, vp, vp, vp, i64, dbl, u64, u64, ci]
        L.rk_tree_create_nd.argtypes = [C.POINTER(vp), ci, ci, ci, C.POINTER(vp), i64, dbl, u64, u64, ci]
        L.rk_tree_destroy.argtypes = [vp]
        L.rk_tree_destroy.restype = None
        L.rk_tree_info.argtypes = [vp, C.POINTER(i64), C.POINTER(dbl)]
        L.rk_tree_get.argtypes = [vp, ci, vp]
        L.rk_tree_nodes.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
        L.rk_tree_state.argtypes = [vp, C.POINTER(vp)]
        L.rk_tree_acc_pot.argtypes = [vp, ci, ci, C.POINTER(vp), dbl, dbl, dbl, C.POINTER(dbl), ci]
        L.rk_tree_exact.argtypes = [vp, ci, ci, i64, dbl, dbl, vp]
        L.rk_tree_cpu_acc_pot.argtypes = [vp, ci, C.POINTER(vp), dbl, dbl, dbl, ci, C.c_uint]
        L.rk_tree_update_particles.argtypes = [vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc:
        raise _EXC.get(rc, RuntimeError)(lib().rk_last_error().decode())
