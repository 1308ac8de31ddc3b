"""Register the operator modules under the import names the reference's callers use, so that
nerf/renderer.py, palette/renderer.py, nerf/network.py, palette/network.py and encoding.py run
unchanged on MI355X:

    import palettenerf_amd.dropin as dropin; dropin.install()
    import raymarching                      # -> palettenerf_amd.raymarching
    from gridencoder import GridEncoder     # -> palettenerf_amd.gridencoder.GridEncoder
    from shencoder import SHEncoder         # -> palettenerf_amd.shencoder.SHEncoder
"""
import sys
import types

from . import gridencoder as _ge
from . import palette_utils as _pu
from . import raymarching as _rm
from . import shencoder as _sh


def _package(name, module, submodule_name):
    pkg = types.ModuleType(name)
    pkg.__path__ = []  # mark as package so `import name.sub` works
    for k, v in vars(module).items():
        if not k.startswith("__"):
            setattr(pkg, k, v)
    setattr(pkg, submodule_name, module)
    sys.modules[name] = pkg
    sys.modules[f"{name}.{submodule_name}"] = module
    return pkg


def install(patch_palette_utils=True):
    """Idempotent.  Returns the dict of installed module names."""
    installed = {
        "raymarching": _package("raymarching", _rm, "raymarching"),         # raymarching/raymarching.py
        "gridencoder": _package("gridencoder", _ge, "grid"),                # gridencoder/grid.py
        "shencoder": _package("shencoder", _sh, "sphere_harmonics"),        # shencoder/sphere_harmonics.py
    }
    if patch_palette_utils and "palette.utils" in sys.modules:
        # palette/utils.py is mostly harness code; only its two HSV operators are native (palette/utils.py:257-295)
        pu = sys.modules["palette.utils"]
        pu.rgb_to_hsv, pu.hsv_to_rgb = _pu.rgb_to_hsv, _pu.hsv_to_rgb
    return installed
