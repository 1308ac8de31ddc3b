"""Build libpnr_hip.so (gfx950) in-tree with hipcc.  `python -m palettenerf_amd.build [--force]`.

One object per .hip translation unit (compiled in parallel), linked into a single shared library
next to this file so that it travels with the source tree.  No torch headers are involved: the
boundary is the plain C ABI in include/pnr.h.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpnr_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: the canonical scalar spec uses explicit fmaf() only.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("PNR_EXTRA_HIPCC_FLAGS", "").split()   # experiment builds only (e.g. -DPNR_MARCH_STATS)
# Per-unit flags.  palette_field: without the SLP vectoriser -- it pairs the epilogue's scalar fp32 math into v_pk_* instructions whose constant operands
# must sit in register PAIRS, hoists those pairs (and a dozen other loop invariants) out of the tile loop and then spills them: 94 -> 10 spilled scalar
# registers, 20 -> 0 bytes of scratch per lane, 162 -> 157 vector registers for the shipped kernel; same arithmetic, same bits (round 5).
UNIT_FLAGS = {"palette_field": ["-fno-slp-vectorize"]}


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: palettenerf_amd needs the ROCm toolchain to build its gfx950 kernels")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hip", ".hpp", ".h", ".inc")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    hdr_m = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc")))
    hdr_m = max(hdr_m, os.path.getmtime(os.path.join(HERE, "..", "include", "pnr.h")))
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_m):
        return obj
    subprocess.check_call([hipcc(), *FLAGS, *UNIT_FLAGS.get(os.path.basename(src)[:-4], []), "-c", src, "-o", obj])
    return obj


def build_variant(name, extra_flags, only=None, verbose=False):
    """An experiment build next to the product library: libpnr_hip_<name>.so from the same sources with extra hipcc flags (-D knobs) on the
    translation units in `only` (default: all), the other objects taken from the product build.  Selected at run time with PNR_LIB_PATH."""
    build()
    vobj = os.path.join(CSRC, "_obj", "variant_" + name)
    os.makedirs(vobj, exist_ok=True)
    objs = []
    for src in sources():
        base = os.path.basename(src)[:-4]
        if only is None or base in only:
            obj = os.path.join(vobj, base + ".o")
            subprocess.check_call([hipcc(), *FLAGS, *UNIT_FLAGS.get(base, []), *extra_flags, "-c", src, "-o", obj])
        else:
            obj = os.path.join(OBJ, base + ".o")
        objs.append(obj)
    lib = os.path.join(HERE, f"libpnr_hip_{name}.so")
    subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib, *objs])
    if verbose:
        print("built", lib)
    return lib


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            if os.path.isfile(os.path.join(OBJ, f)):
                os.remove(os.path.join(OBJ, f))
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, sources()))
    subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs])
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:      # python -m palettenerf_amd.build --variant NAME [--only unit,unit] -- -DFLAG ...
        i = sys.argv.index("--variant")
        only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
        flags = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
        build_variant(sys.argv[i + 1], flags, only, verbose=True)
    else:
        build(force="--force" in sys.argv, verbose=True)
