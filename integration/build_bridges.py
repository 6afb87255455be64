"""Compiles the two reference-side bindings (integration/rakau_amd_bridge.cpp: the ROCm seam, rocm_fwd.hpp:22-46;
integration/rakau_amd_cuda_bridge.cpp: the CUDA seam, cuda_fwd.hpp:23-30) against the reference's own HEADERS and links the
two C++ drivers of tests/cpp against them -> tests/build/. Needs a checkout of the reference (headers only, read at compile
time; nothing of it is copied). Used by __graft_entry__.build() and tests/test_integration_bridge.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"
BUILD = os.path.join(ROOT, "tests", "build")
LIB = os.path.join(BUILD, "librakau_rocm_bridge.so")
DRIVER = os.path.join(BUILD, "bridge_driver")
CUDA_LIB = os.path.join(BUILD, "librakau_cuda_bridge.so")
CUDA_DRIVER = os.path.join(BUILD, "cuda_bridge_driver")
RK_LIBDIR = os.path.join(ROOT, "rakau_amd", "lib")
SOURCES = [os.path.join(ROOT, "integration", f) for f in ("rakau_amd_bridge.cpp", "rakau_amd_cuda_bridge.cpp", "rakau_amd_bridge.hpp",
                                                           "rakau_amd_cuda_bridge.hpp", "rakau_amd_bridge_common.hpp")] + \
          [os.path.join(ROOT, "tests", "cpp", f) for f in ("bridge_driver.cpp", "cuda_bridge_driver.cpp")] + \
          [os.path.join(ROOT, "include", "rakau_amd.h")]


def have_reference():
    return os.path.isdir(REF_INC)


def up_to_date():
    outs = (LIB, DRIVER, CUDA_LIB, CUDA_DRIVER)
    if not all(os.path.exists(o) for o in outs):
        return False
    newest_src = max(os.path.getmtime(s) for s in SOURCES if os.path.exists(s))
    return min(os.path.getmtime(o) for o in outs) >= newest_src


def build(force=True):
    if not force and up_to_date():
        return
    os.makedirs(BUILD, exist_ok=True)
    common = ["g++", "-std=c++17", "-O2", "-fPIC", "-Wall", "-Wextra", "-Wno-comment", "-I" + REF_INC, "-I" + os.path.join(ROOT, "include")]
    subprocess.check_call(common + ["-shared", os.path.join(ROOT, "integration", "rakau_amd_bridge.cpp"), "-L" + RK_LIBDIR,
                                    "-lrakau_amd", "-Wl,-rpath," + RK_LIBDIR, "-o", LIB])
    subprocess.check_call(common + ["-pthread", os.path.join(ROOT, "tests", "cpp", "bridge_driver.cpp"), "-L" + BUILD, "-lrakau_rocm_bridge",
                                    "-L" + RK_LIBDIR, "-lrakau_amd", "-Wl,-rpath," + BUILD, "-Wl,-rpath," + RK_LIBDIR, "-o", DRIVER])
    # The CUDA seam (include/rakau/detail/cuda_fwd.hpp:23-30): the reference's multi-GPU entry.
    subprocess.check_call(common + ["-shared", "-pthread", os.path.join(ROOT, "integration", "rakau_amd_cuda_bridge.cpp"), "-L" + RK_LIBDIR,
                                    "-lrakau_amd", "-Wl,-rpath," + RK_LIBDIR, "-o", CUDA_LIB])
    subprocess.check_call(common + ["-pthread", os.path.join(ROOT, "tests", "cpp", "cuda_bridge_driver.cpp"), "-L" + BUILD,
                                    "-lrakau_cuda_bridge", "-L" + RK_LIBDIR, "-lrakau_amd", "-Wl,-rpath," + BUILD, "-Wl,-rpath," + RK_LIBDIR,
                                    "-o", CUDA_DRIVER])


if __name__ == "__main__":
    build()
