#!/usr/bin/env python3
"""Builds oracle/_ref/ -- the part of the REFERENCE that compiles here from its own sources, where they lie (TEST INFRASTRUCTURE).

Two builds:

1. `build()` (CPU, runs anywhere): `palette/src/bindings.cpp` (plain C++ / pybind11; holds compute_RGB_histogram, bindings.cpp:40-91), compiled
   UNMODIFIED from /root/reference with g++ against the torch / pybind11 headers of this image.  The two symbols it takes from palette.cu
   (rgb_to_hsv / hsv_to_rgb, palette/src/palette_func.h) come from this repo's PRODUCT binding of the C ABI for that extension
   (palettenerf_amd/csrc/shim/palette_func_hip.cpp -> libpnr_hip.so): reference pybind layer + reference histogram code + this repo's kernels.
2. `build_hip()` (round 4): the reference's CUDA extensions themselves -- raymarching/src/{raymarching.cu,bindings.cpp},
   shencoder/src/{shencoder.cu,bindings.cpp}, palette/src/{palette.cu,bindings.cpp} -- compiled UNMODIFIED for gfx950 with
   torch.utils.cpp_extension (torch's own hipify pass + hipcc; no header, library or tool is stood in for).  The sources are read where they lie;
   hipify writes its translated copies into a scratch directory under /tmp; only the resulting ref_<name>.so files are copied into oracle/_ref/.
   gridencoder/src/gridencoder.cu does NOT compile (atomicAdd(__half2*, __half2) has no overload in ROCm 7.2's HIP headers) and is left unbuilt.
   tests/test_gpu_reference_kernels.py and profiles/reference_kernels.py run these modules on the MI355X next to this repository's kernels.

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


# ------------------------------------------------------------------------------------------------------------------------------------------
# Round 4: the reference's CUDA extensions themselves, compiled for gfx950 by the image's own toolchain for CUDA extensions on ROCm.
#
# torch.utils.cpp_extension (what the reference's */backend.py call) translates a .cu with torch.utils.hipify and compiles it with hipcc -- that is
# how ANY CUDA extension is built on a ROCm PyTorch, no file of this repository is involved.  The reference's own recipe fails here only because
# it passes -std=c++14 (raymarching/backend.py:7,12) to a torch 2.10 whose headers need C++17; with -std=c++17 (and the HIP spelling of its
# -U__CUDA_NO_HALF_* flags) three of the four extensions compile UNMODIFIED from /root/reference:
#     _raymarching (raymarching.cu: all 14 kernels), _shencoder (shencoder.cu), _palette_func (palette.cu + bindings.cpp).
# _gridencoder does NOT: gridencoder.cu:269 calls atomicAdd(__half2*, __half2), an overload ROCm 7.2's headers do not have (they offer
# unsafeAtomicAdd); supplying it would be a stand-in for a library function the image lacks, so that extension stays unbuilt and the hash
# grid stays pinned by the reference's Python wrapper over the oracle (oracle/native_facade.py) and by the independent formulations.
# hipify writes its translation next to the input, /root/reference is read-only: the sources are copied to a scratch directory OUTSIDE the
# repository, built there, and only the .so files are kept (oracle/_ref/ref_<name>.so).  On the GPU box the tests load them and run the
# reference's own kernels on the MI355X next to this repository's (tests/test_gpu_reference_kernels.py); bench.py times them (extra.reference_kernels).
HIP_EXTENSIONS = {
    "raymarching": ("_raymarching", ("raymarching.cu", "bindings.cpp")),
    "shencoder": ("_shencoder", ("shencoder.cu", "bindings.cpp")),
    "palette": ("_palette_func", ("palette.cu", "bindings.cpp")),
}
REF_ROOT = "/root/reference"


def hip_path(ext):
    return os.path.join(OUT_DIR, f"ref_{ext}.so")


def hip_available(ext):
    return os.path.exists(hip_path(ext))


def build_hip(force=False, verbose=False):
    """Compile the three buildable reference extensions for gfx950 (no GPU needed).  Returns {ext: path or None}."""
    out = {}
    for ext, (modname, files) in HIP_EXTENSIONS.items():
        srcs = [os.path.join(REF_ROOT, ext, "src", f) for f in files]
        dst = hip_path(ext)
        if not all(os.path.exists(p) for p in srcs):
            out[ext] = dst if os.path.exists(dst) else None
            continue
        hdrs = [os.path.join(REF_ROOT, ext, "src", f) for f in os.listdir(os.path.join(REF_ROOT, ext, "src")) if f.endswith(".h")]
        if not force and os.path.exists(dst) and os.path.getmtime(dst) >= max(os.path.getmtime(p) for p in srcs + hdrs + [os.path.abspath(__file__)]):
            out[ext] = dst
            continue
        import shutil
        import tempfile
        os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
        os.environ.setdefault("MAX_JOBS", "4")
        from torch.utils.cpp_extension import load
        tmp = tempfile.mkdtemp(prefix="pnr_refbuild_", dir="/tmp")
        try:
            shutil.copytree(os.path.join(REF_ROOT, ext, "src"), os.path.join(tmp, "src"))      # scratch copy, outside the repository: hipify writes next to its input
            bd = os.path.join(tmp, "build")
            os.makedirs(bd)
            half = ["-U__CUDA_NO_HALF_OPERATORS__", "-U__CUDA_NO_HALF_CONVERSIONS__", "-U__CUDA_NO_HALF2_OPERATORS__"]       # the reference's own flags (backend.py:8) ...
            half += [f.replace("CUDA", "HIP") for f in half]                                                              # ... and their HIP spelling
            load(name=modname, extra_cflags=["-O3", "-std=c++17"], extra_cuda_cflags=["-O3", "-std=c++17", *half],
                 sources=[os.path.join(tmp, "src", f) for f in files], build_directory=bd, verbose=verbose, is_python_module=False)
            os.makedirs(OUT_DIR, exist_ok=True)
            shutil.copy2(os.path.join(bd, modname + ".so"), dst)
            out[ext] = dst
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def load_hip(ext):
    """Import oracle/_ref/ref_<ext>.so as the reference's extension module (its own PyInit name); needs torch imported (libtorch symbols)."""
    import importlib.util
    import torch  # noqa: F401
    modname = HIP_EXTENSIONS[ext][0]
    spec = importlib.util.spec_from_file_location(modname, hip_path(ext))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


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
    print(build_hip(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
