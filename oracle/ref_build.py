#!/usr/bin/env python3
"""Builds oracle/_ref/ -- the part of the REFERENCE that compiles here from its own sources, where they lie (TEST INFRASTRUCTURE).

What is buildable: `palette/src/bindings.cpp` (plain C++ / pybind11; holds compute_RGB_histogram, bindings.cpp:40-91).  It is compiled
UNMODIFIED from /root/reference with g++ against the torch / pybind11 headers of this image.  The reference links it with palette.cu (CUDA:
no nvcc, no CUDA headers here -> unbuildable, and no stand-in is written for it); the two symbols bindings.cpp takes from palette.cu
(rgb_to_hsv / hsv_to_rgb, palette/src/palette_func.h) are provided by this repo's PRODUCT binding of the C ABI for that extension
(palettenerf_amd/csrc/shim/palette_func_hip.cpp -> libpnr_hip.so), i.e. the module is the reference's `_palette_func` as it would be built on an
MI355X box: reference pybind layer + reference histogram code + this repo's HIP kernels.  Everything else under /root/reference on the
hot path is CUDA (raymarching.cu, gridencoder.cu, shencoder.cu, palette.cu) and stays unbuilt.

Use: tests/golden/gen_golden.py imports the module to write tests/golden/hist.npz; tests/test_oracle.py checks the oracle's
compute_RGB_histogram against the module directly when it is present.  Outputs only into oracle/_ref/ (git-ignored, travels to the GPU box).
Nothing here runs on the GPU box: /root/reference does not exist there; the prebuilt .so is used as it is.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
REF_SRC = "/root/reference/palette/src/bindings.cpp"
SHIM = os.path.join(ROOT, "palettenerf_amd", "csrc", "shim", "palette_func_hip.cpp")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "_palette_func" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def available():
    return os.path.exists(OUT)


def build(force=False, verbose=False):
    """Compile if /root/reference is present and the output is stale; returns the module path or None (no reference here, nothing prebuilt)."""
    if not os.path.exists(REF_SRC):
        return OUT if available() else None
    lib = os.path.join(ROOT, "palettenerf_amd", "libpnr_hip.so")
    deps = [REF_SRC, SHIM, os.path.join(ROOT, "include", "pnr.h")]
    if not force and available() and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    if not os.path.exists(lib):
        raise RuntimeError("build libpnr_hip.so first (python -m palettenerf_amd.build): the reference's _palette_func links against it")
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}"]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=_palette_func", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w", *inc, REF_SRC, SHIM, "-o", OUT,
           f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", f"-L{os.path.dirname(lib)}", "-l:libpnr_hip.so",
           f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN/../../palettenerf_amd"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


def load():
    """Import the built module (needs torch imported first for its symbols)."""
    import importlib.util
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location("_palette_func", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
