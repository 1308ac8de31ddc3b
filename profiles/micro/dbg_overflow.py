import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, ctypes
from palettenerf_amd import network, raymarching, scene, fused
cuda = torch.device("cuda:0")
m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=100.0, min_near=0.2)
scene.seed_field_(m, 11)
with torch.no_grad():
    m.color_net[0].weight[5].mul_(4e6); m.color_net[1].weight[:, 5].mul_(1/4e6)
m = m.to(cuda).eval()
m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
pose = torch.from_numpy(scene.lookat_pose())[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(48, 48), 48, 48)
ro, rd = ro.to(cuda), rd.to(cuda)
m.march_mode, m.fused_field = "native", True
m._fused = fused.NeRFFieldFused(m)
f = m._fused
print("plan", f.frame_precision(), f._guard()[1])
orig = f._note_overflow
f._note_overflow = lambda: (print("NOTE OVERFLOW CALLED"), orig())
with torch.no_grad():
    r = m.render(ro, rd, perturb=False, dt_gamma=0.0, max_steps=1024, T_thresh=1e-4)
print("after", f.frame_precision(), r.get("iterations"), torch.isfinite(r["image"]).all().item())
