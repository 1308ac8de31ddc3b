"""Experiment: one 800x800 frame as K independent ray subsets rendered concurrently (one host thread + HIP stream each) versus
one call.  Measures wall time per frame; images are identical by construction (per-ray results do not depend on the batch)."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from palettenerf_amd import raymarching, scene
from palettenerf_amd.fused import NeRFFieldFused, tile_ray_order

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
H = W = 800
pose = torch.from_numpy(scene.lookat_pose())[None]
ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
order = tile_ray_order(torch.arange(H * W), W, 8).long()
ro, rd = ro[0][order].contiguous().to(dev), rd[0][order].contiguous().to(dev)      # rays physically in tile order
nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_infer, m.min_near)
N = ro.shape[0]
full = NeRFFieldFused(m)
parts = []
for k in range(K):
    lo, hi = N * k // K, N * (k + 1) // K
    parts.append((NeRFFieldFused(m), torch.cuda.Stream(), ro[lo:hi].contiguous(), rd[lo:hi].contiguous(), nears[lo:hi].contiguous(), fars[lo:hi].contiguous()))


def one():
    return full.render_frame(ro, rd, nears, fars, 0.0, 1024, 1e-4)


def split():
    outs = [None] * K

    def work(k):
        f, st, o, d, n, fa = parts[k]
        with torch.cuda.stream(st):
            outs[k] = f.render_frame(o, d, n, fa, 0.0, 1024, 1e-4)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return outs


with torch.no_grad():
    for fn, name in ((one, "one call"), (split, f"{K} concurrent subsets")):
        for _ in range(4):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        rendered = r[3]["rendered"] if name == "one call" else sum(x[3]["rendered"] for x in r)
        print(f"{name}: {dt:.3f} ms/frame, rendered {rendered}")
    a = one()
    b = split()
    img = torch.cat([x[2] for x in b])
    print("max |image diff| =", float((a[2] - img).abs().max()))
