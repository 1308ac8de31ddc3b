import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from palettenerf_amd import network, raymarching, renderer, scene
from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused
from palettenerf_amd.pipeline import FramesInFlight
dev = torch.device("cuda", 0)
for kind in ("nerf", "palette"):
    m = network.NeRFNetwork(bound=2, cuda_ray=True) if kind == "nerf" else network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True)
    scene.seed_field_(m, 0)
    m = m.to(dev).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(dev))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.march_mode, m.fused_field, m.count_rendered = "native", True, True
    m._fused = NeRFFieldFused(m) if kind == "nerf" else PaletteFieldFused(m)
    H = W = 256
    intr = scene.intrinsics_from_fov(H, W)
    rays = []
    for i in range(24):
        pose = torch.from_numpy(scene.lookat_pose(azimuth_deg=11.0 * i))[None]
        ro, rd = scene.get_rays(pose, intr, H, W)
        rays.append((ro.to(dev), rd.to(dev)))
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    if kind == "palette":
        kw["gui_mode"] = False
    with torch.no_grad():
        want = [m.render(ro, rd, **kw)["image"].clone() for ro, rd in rays]
    bad = 0
    for F in (2, 3, 4):
        fif = FramesInFlight(m, F)
        for rep in range(8):
            got = fif.render(lambda i: rays[i % 24], 96, consume=lambda i, r: r["image"].clone(), **kw)
            for i, g in enumerate(got):
                a, b = g.cpu().numpy(), want[i % 24].cpu().numpy()
                if not np.array_equal(np.nan_to_num(a), np.nan_to_num(b)):
                    bad += 1
    print(kind, "frames compared:", 3 * 8 * 96, "mismatching:", bad)
