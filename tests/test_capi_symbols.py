"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re

from rakau_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for hdr in ("rakau_amd.h", "rakau_amd_tree.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"RK_EXPORT\s+[\w\s\*]+?\b(rk_\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 20
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing
    # The Python binding lists exactly the declared symbols.
    assert sorted(_capi.SYMBOLS) == decl


def test_no_gpu_queries_do_not_crash():
    lib = _capi.lib()
    assert lib.rk_min_size() == 64
    assert lib.rk_device_count() >= 0
    assert lib.rk_has_accelerator() in (0, 1)


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "rakau_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(base, f)
    for f in ("rakau_amd.h", "rakau_amd_tree.h", os.path.join("rakau_amd", "tree.hpp"),
              os.path.join("rakau_amd", "kwargs.hpp"), os.path.join("rakau_amd", "cpu_engine.hpp")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower()


def test_cross_check_kernels_live_in_their_own_library():
    """The kernels that were measured and lost (variant 1: k_dfs_wave / k_dfs_block; variant 4: k_lists / k_dense / k_combine)
    are not in librakau_amd.so: they build into librakau_amd_xcheck.so, which exports the one entry the product binds on demand
    (rk_xcheck.hpp). The product library stays below 8 MB (round 3: 20.7)."""
    libdir = os.path.dirname(_capi.LIB_PATH)
    product = open(_capi.LIB_PATH, "rb").read()
    xpath = os.path.join(libdir, "librakau_amd_xcheck.so")
    assert os.path.exists(xpath)
    xcheck = open(xpath, "rb").read()
    for kernel in (b"k_dfs_wave", b"k_dfs_block", b"k_lists", b"k_dense", b"k_combine"):
        assert kernel not in product, kernel
        assert kernel in xcheck, kernel
    for kernel in (b"k_list", b"k_pc", b"k_super", b"k_census"):
        assert kernel in product, kernel
    assert hasattr(ctypes.CDLL(xpath), "rk_xcheck_entry")
    if "RAKAU_AMD_LIB" not in os.environ:
        assert len(product) < 8 * 2 ** 20, len(product)


def test_selecting_a_cross_check_variant_without_a_state_is_an_error_not_a_crash():
    lib = _capi.lib()
    assert lib.rk_set_kernel_variant(None, 1) != 0
    assert b"variant" in lib.rk_last_error()


def test_environment_knobs_are_few_and_documented():
    """The library reads at most 25 RK_* environment variables, and INTEGRATION.md's table names every one."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    knobs = set()
    for f in glob.glob(os.path.join(root, "rakau_amd", "csrc", "*")) + glob.glob(os.path.join(root, "include", "**", "*.h*"), recursive=True):
        if os.path.isfile(f):
            knobs.update(re.findall(r'getenv\("(RK_[A-Z0-9_]+)"\)', open(f, errors="replace").read()))
    assert 0 < len(knobs) <= 25, sorted(knobs)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [k for k in sorted(knobs) if "`%s`" % k not in doc]
    assert not missing, missing
