"""Where the host time of one native NeRF frame goes: cProfile over 50 m.render() calls (GPU time is inside pnr_nerf_render_frame)."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from palettenerf_amd import scene
from palettenerf_amd.fused import NeRFFieldFused, tile_ray_order

sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
m.march_mode, m.fused_field = "native", True
m._fused = NeRFFieldFused(m)
H = W = 800
pose = torch.from_numpy(scene.lookat_pose())[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
ro, rd = ro.to(dev), rd.to(dev)
m._fused.ray_order = tile_ray_order(torch.arange(H * W), W, 8).to(dev)
kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
with torch.no_grad():
    for _ in range(5):
        m.render(ro, rd, **kw)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        m.render(ro, rd, **kw)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
