import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from palettenerf_amd import _lib, network, raymarching, renderer, scene
from palettenerf_amd.fused import PaletteFieldFused, tile_ray_order
cuda = torch.device("cuda:0")
m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
scene.seed_field_(m, 5)
m = m.to(cuda).eval()
m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
m.count_rendered = True
m.march_mode, m.fused_field = "native", True
H, W = 40, 56
pose = torch.from_numpy(scene.lookat_pose())[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
ro, rd = ro.to(cuda), rd.to(cuda)
m._fused = PaletteFieldFused(m)
kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4, gui_mode=False)
lib = _lib.load()
res = {}
with torch.no_grad():
    for w12 in (0, 1):
        lib.pnr_set_option(b"palette_waves12", w12)
        for name, order in (("none", None), ("tile", tile_ray_order(torch.arange(H * W), W, 8)), ("rand", torch.randperm(H * W, generator=torch.Generator().manual_seed(1)).to(torch.int32))):
            m._fused.ray_order = None if order is None else order.to(cuda)
            res[(w12, name)] = m.render(ro, rd, **kw)
    m.march_mode, m.fused_field = "compat", False
    ref = m.render(ro, rd, **kw)
for k, r in res.items():
    print(k, int(r["rendered"].sum()), {n: float((r[n] - ref[n]).abs().max()) for n in ("image", "basis_rgb", "weights_sum")}, "vs (0,none):", {n: float((r[n] - res[(0, 'none')][n]).abs().max()) for n in ("image", "basis_rgb")})
