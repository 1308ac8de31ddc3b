#!/usr/bin/env python3
"""where do the 3e-4 (edited image), 2e-4 (depth) and 5e-4 (depth_origin) tolerances of the frame tests come from?
For every golden with an edit: max abs error, number of pixels above 1e-4 / 5e-5 / 2e-5, in the host-driven mirror (compat: torch ops on the
HIP operators, fp32 everywhere), the device-driven loop (split-fp16 field) and the device-driven loop with the exact-fp32 field; and the same
for depth / depth_origin.  Also how far the two GPU paths are from EACH OTHER (is the golden or the kernel the outlier?)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from palettenerf_amd import network, raymarching, renderer, scene  # noqa: E402
from palettenerf_amd.fused import PaletteFieldFused  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
dev = torch.device("cuda:0")


def stats(name, got, want):
    got = got.detach().cpu().numpy().astype(np.float64)
    m = np.isfinite(want)
    e = np.abs(got - want)[m]
    per_px = e.reshape(-1, want.shape[-1]).max(-1) if want.ndim > 1 and e.size == want.size else e
    print(f"    {name:34s} max {e.max():.2e}  pixels > 1e-4: {(per_px > 1e-4).sum():4d}  > 5e-5: {(per_px > 5e-5).sum():4d}  > 2e-5: {(per_px > 2e-5).sum():4d}  of {per_px.size}  (|value| max {np.abs(want[m]).max():.2f})")
    return e.max()


for case in ("a", "b"):
    g = np.load(os.path.join(GOLDEN, f"frame_palette_{case}.npz"))
    opt = renderer.default_opt(pred_clip=bool(g["pred_clip"]))
    m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(dev).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(dev))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    H, W = int(g["H"]), int(g["W"])
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro.to(dev), rd.to(dev)
    kw = dict(dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4)
    print(f"frame_palette_{case}: {H}x{W}, dt_gamma {float(g['dt_gamma'])}, density_scale {float(g['density_scale'])}")
    outs = {}
    for mode in ("compat", "native", "native_fp32"):
        m.march_mode = "native" if mode.startswith("native") else "compat"
        m.fused_field = mode != "compat"
        if m.fused_field:
            m._fused = PaletteFieldFused(m)
            m._fused.precision = 0 if mode == "native_fp32" else 1
        m.edit = None
        with torch.no_grad():
            r = m.render(ro, rd, gui_mode=False, **kw)
        print(f"  {mode}: unedited")
        stats("image", r["image"], g["image"])
        stats("depth", r["depth"], g["depth"])
        stats("depth_origin", r["depth_origin"], g["depth_origin"])
        m.edit = renderer.RegionEdit(opt)
        m.edit.update_cent(mean_xyz=torch.tensor([0.1, 0.0, -0.2], device=dev))
        m.edit.update_std(std_xyz=0.5)
        m.edit.update_delta_hsv(m.basis_color.data.clamp(0, 1), (m.basis_color.data * 0.6 + 0.2).flip(0).clamp(0, 1))
        print("    delta_hsv (hue shift, saturation gain, value gain per basis):", [[round(float(v), 3) for v in row] for row in m.edit.delta_hsv])
        with torch.no_grad():
            r2 = m.render(ro, rd, gui_mode=True, **kw)
        print(f"  {mode}: RegionEdit active")
        stats("edit_image", r2["image"], g["edit_image"])
        outs[mode] = r2["image"].detach().cpu().numpy()
    for a, b in (("compat", "native"), ("native", "native_fp32")):
        d = np.abs(outs[a] - outs[b])
        print(f"  edited image, {a} vs {b}: max {np.nanmax(d):.2e}, pixels > 1e-4: {(np.nanmax(d.reshape(-1, 3), -1) > 1e-4).sum()}")
