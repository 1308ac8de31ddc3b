#!/usr/bin/env python3
"""A/B of the training march's counting pass: cooperative (train_coop 1) vs one ray per lane (0), same inputs -> same rows; timings."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from palettenerf_amd import raymarching, scene, _lib
dev = torch.device("cuda:0")
lib = _lib.load()

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3

def rig_rays(n, seed):
    H, W = 756, 1008
    p = np.eye(4, dtype=np.float32); p[:3, 0], p[:3, 1], p[:3, 2] = [1, 0, 0], [0, -1, 0], [0, 0, -1]; p[:3, 3] = [0.3, 0.0, 1.5]
    ro, rd = scene.get_rays(torch.from_numpy(p[None]), scene.intrinsics_from_fov(H, W, 0.9), H, W)
    g = torch.Generator().manual_seed(seed); idx = torch.randint(0, H * W, [n], generator=g)
    return ro[0, idx].contiguous().to(dev), rd[0, idx].contiguous().to(dev)

def lego_rays(n, seed):
    H = W = 800
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose())[None], scene.intrinsics_from_fov(H, W), H, W)
    g = torch.Generator().manual_seed(seed); idx = torch.randint(0, H * W, [n], generator=g)
    return ro[0, idx].contiguous().to(dev), rd[0, idx].contiguous().to(dev)

out = {}
for name, grid_np, rays_fn, dtg, min_near in (("slab_dt1/128", scene.slab_density_grid(), rig_rays, 1 / 128, 0.02), ("lego_dt0", scene.brick_density_grid(), lego_rays, 0.0, 0.2),
                                               ("lego_dt1/128", scene.brick_density_grid(), lego_rays, 1 / 128, 0.2), ("slab_dt0", scene.slab_density_grid(), rig_rays, 0.0, 0.02)):
    grid = torch.from_numpy(grid_np).to(dev)
    bits = raymarching.packbits(grid, 0.5)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2], dtype=torch.float32, device=dev)
    for n in (4096, 40000):
        for seed in (0, 1):
            ro, rd = rays_fn(n, seed)
            nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, min_near)
            res = {}
            for coop in (0, 1):
                lib.pnr_set_option(b"train_coop", coop)
                torch.manual_seed(seed)
                cnt = torch.zeros(2, dtype=torch.int32, device=dev)
                x, d, dl, rays = raymarching.march_rays_train(ro, rd, 2.0, bits, 2, 128, nears, fars, cnt, -1, True, 128, True, dtg, 1024)
                res[coop] = (x.clone(), d.clone(), dl.clone(), rays.clone(), cnt.clone())
            same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
            rec = {"same": bool(same), "samples": int(res[0][4][0]), "samples_coop": int(res[1][4][0])}
            if not same:
                r0, r1 = res[0][3], res[1][3]
                bad = (r0[:, 2] != r1[:, 2]).nonzero().flatten()
                rec["rays_with_different_counts"] = int(bad.numel())
                if bad.numel(): rec["first"] = [int(bad[0]), r0[bad[0]].tolist(), r1[bad[0]].tolist()]
            if seed == 0:
                for coop in (0, 1):
                    lib.pnr_set_option(b"train_coop", coop)
                    def f():
                        cnt = torch.zeros(2, dtype=torch.int32, device=dev)
                        raymarching.march_rays_train(ro, rd, 2.0, bits, 2, 128, nears, fars, cnt, -1, True, 128, True, dtg, 1024)
                    rec[f"us_incl_wrapper_coop{coop}"] = timed(f)
            out[f"{name}_n{n}_seed{seed}"] = rec
            print(name, n, seed, rec, flush=True)
lib.pnr_set_option(b"train_coop", 1)
print(json.dumps(out))
