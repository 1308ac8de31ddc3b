"""Is the first lookup launch of a frame slow because the hash table has been evicted from the caches by the previous frame's streaming traffic?
Renders frames of the bench camera path; before every second one the table is read once (emb.sum()).  Run under rocprofv3 --kernel-trace and
compare the first k_frame_grid launch after each k_frame_init (profiles/kernel_quantiles.py prints launches in order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from palettenerf_amd import scene

sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
m.march_mode = "native"
H = W = 800
kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
rays = []
for i in range(12):
    pose = torch.from_numpy(bench.pose_of(args, i))[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
    rays.append((ro.to(dev), rd.to(dev)))
emb = m.encoder.embeddings
with torch.no_grad():
    for i in range(4):
        m.render(*rays[i], **kw)
    torch.cuda.synchronize()
    for i in range(4, 12):
        if i % 2 == 1:
            s = emb.sum()      # marks the frames whose table was just read: a reduce kernel in front of k_frame_init
        m.render(*rays[i], **kw)
    torch.cuda.synchronize()
