"""palettenerf_amd -- MI355X (gfx950) native volumetric-rendering operators for PaletteNeRF.

The package mirrors the reference's operator API for the hot path only:

    palettenerf_amd.raymarching    <->  raymarching/raymarching.py
    palettenerf_amd.gridencoder    <->  gridencoder/grid.py
    palettenerf_amd.shencoder      <->  shencoder/sphere_harmonics.py
    palettenerf_amd.palette_utils  <->  palette/utils.py (rgb_to_hsv / hsv_to_rgb)

all backed by hand-written HIP kernels behind the C ABI in include/pnr.h (libpnr_hip.so).
`palettenerf_amd.dropin.install()` registers those modules under the reference's import names so
that nerf/renderer.py and palette/renderer.py run unchanged.
"""
__version__ = "0.1.0"
