#!/usr/bin/env python3
"""How many samples of a configs[3]-shaped training step receive an exactly-zero gradient (samples behind the point where a ray's transmittance
fell below T_thresh: composite_rays_train stops there, raymarching.cu:455-459)?  Per sample and per 32-sample tile."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from palettenerf_amd import raymarching as rm
dev = torch.device("cuda:0")
for kind in ("nerf", "palette"):
    m, step = bench.make_training_step(kind, 4096, dev)
    for i in range(30):
        step(i)
    stats = {}
    orig = rm.composite_rays_train

    class Obs(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t, name):
            ctx.name = name
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            z = (g == 0) if g.ndim == 1 else (g == 0).all(dim=-1)
            n = z.numel() // 32 * 32
            stats[ctx.name + "_zero_frac"] = float(z.float().mean())
            stats[ctx.name + "_zero_tiles_frac"] = float(z[:n].view(-1, 32).all(dim=1).float().mean())
            return g, None

    def wrapped(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        if sigmas.requires_grad:
            sigmas = Obs.apply(sigmas, "sigma")
        if rgbs.requires_grad:
            rgbs = Obs.apply(rgbs, "rgb")
        return orig(sigmas, rgbs, deltas, rays, T_thresh)
    rm.composite_rays_train = wrapped
    try:
        step(31)
        torch.cuda.synchronize()
    finally:
        rm.composite_rays_train = orig
    print(kind, stats, flush=True)
