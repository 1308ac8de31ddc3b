"""Cost of the hash-grid lookup per level: time pnr_grid_encode_forward on the first-hit samples of the S0 800x800 frame
(tile-ordered rays, the coherence the frame loop sees) with L = 1..16 levels; the increments are the per-level costs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from palettenerf_amd import raymarching, scene
import ctypes
from palettenerf_amd._torch_glue import call, ptr, require
_u32, _f32, _int = ctypes.c_uint32, ctypes.c_float, ctypes.c_int
from palettenerf_amd.fused import tile_ray_order

sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
H = W = 800
pose = torch.from_numpy(scene.lookat_pose())[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
order = tile_ray_order(torch.arange(H * W), W, 8).long() if "--rowmajor" not in sys.argv else torch.arange(H * W)
ro, rd = ro[0][order].contiguous().to(dev), rd[0][order].contiguous().to(dev)
nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_infer, m.min_near)
N = ro.shape[0]
alive = torch.arange(N, dtype=torch.int32, device=dev)
xyzs, dirs, deltas = raymarching.march_rays(N, 1, alive, nears.clone(), ro, rd, m.bound, m.density_bitfield, m.cascade, m.grid_size, nears, fars, -1, False, 0.0, 1024)
keep = deltas[:, 0] > 0
x = ((xyzs[keep] + m.bound) / (2 * m.bound)).contiguous()
B = x.shape[0]
enc = m.encoder
emb = enc.embeddings.detach()
print("samples", B, "table rows", emb.shape[0])
out = torch.empty(16, B, 2, device=dev)


def run(L):
    call("pnr_grid_encode_forward", ptr(x), ptr(emb), ptr(enc.offsets), ptr(out), _u32(B), _u32(3), _u32(2), _u32(L), _f32(np.log2(enc.per_level_scale)),
         _u32(enc.base_resolution), None, _u32(enc.gridtype_id), _int(int(enc.align_corners)), _int(0))


prev = 0.0
offs = enc.offsets.cpu().numpy()
for L in range(1, 17):
    for _ in range(3):
        run(L)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        run(L)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    rows = int(offs[L] - offs[L - 1])
    print(f"L={L:2d} total {t:7.1f} us  level {L - 1:2d}: +{t - prev:6.1f} us   table {rows * 8 / 1e6:6.2f} MB  hashed={rows == 524288}")
    prev = t
