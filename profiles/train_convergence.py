#!/usr/bin/env python3
"""End-to-end training check on synthetic data (no datasets offline): a teacher field (seeded NeRF, smooth: only the coarse levels of
its table are non-zero; scene S0 occupancy) is rendered from a ring of cameras with the native frame loop; a student with the
reference's default initialisation is trained on those images exactly as the reference trainer does it -- random pixels per step
(on-device get_rays), march_rays_train, field, composite_rays_train, MSE on RGB, Adam (lr 1e-2, betas (0.9, 0.99), eps 1e-15),
update_extra_state every 16 steps -- and evaluated on a held-out view.  Prints PSNR over the steps.
Exercises every training kernel: perturbed march with the occupancy mip, hash-grid forward, binned table gradient, MFMA weight
gradient, composite forward/backward, occupancy maintenance."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from palettenerf_amd import network, raymarching, rays, scene  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--res", type=int, default=200)
    ap.add_argument("--views", type=int, default=24)
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    H = W = args.res
    intr = scene.intrinsics_from_fov(H, W)
    poses = torch.from_numpy(np.stack([scene.lookat_pose(elevation_deg=20.0 + 25.0 * (i % 3), azimuth_deg=360.0 * i / args.views) for i in range(args.views + 1)])).to(dev)

    teacher = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=25.0, min_near=0.2)
    scene.seed_field_(teacher, 0)
    with torch.no_grad():  # smooth teacher: keep levels 0..5, silence the finer ones
        off = teacher.encoder.offsets
        teacher.encoder.embeddings[int(off[6]):] = 0
    teacher = teacher.to(dev).eval()
    teacher.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(dev))
    raymarching.packbits(teacher.density_grid, 0.5, teacher.density_bitfield)
    teacher.march_mode, teacher.fused_field = "native", True
    images = []
    with torch.no_grad():
        for p in poses:
            r = rays.get_rays(p[None], intr, H, W, -1)
            images.append(teacher.render(r["rays_o"], r["rays_d"], perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4, bg_color=1)["image"][0])
    images = torch.stack(images)                       # [V+1, H*W, 3]; the last view is held out
    print(f"teacher: {args.views} training views + 1 held-out, {H}x{W}, mean colour {images.mean(dim=(0, 1)).tolist()}")

    student = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=1.0, min_near=0.2).to(dev)
    opt = torch.optim.Adam(student.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)

    def evaluate():
        student.eval()
        student.march_mode, student.fused_field = "native", True
        with torch.no_grad():
            r = rays.get_rays(poses[-1:], intr, H, W, -1)
            img = student.render(r["rays_o"], r["rays_d"], perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4, bg_color=1)["image"][0]
        student.train()
        return scene.psnr(img, images[-1])

    student.train()
    t0 = time.perf_counter()
    log = []
    for step in range(args.steps + 1):
        if step % 16 == 0:
            with torch.no_grad():
                student.update_extra_state()
        if step % 250 == 0:
            torch.cuda.synchronize()
            log.append((step, evaluate(), time.perf_counter() - t0))
            print(f"step {step:5d}  held-out PSNR {log[-1][1]:6.2f} dB   {log[-1][2]:6.1f} s")
        v = int(torch.randint(0, args.views, (1,)))
        r = rays.get_rays(poses[v:v + 1], intr, H, W, args.rays)
        target = images[v][r["inds"][0]]
        out = student.render(r["rays_o"], r["rays_d"], perturb=True, dt_gamma=0, max_steps=1024, T_thresh=1e-4, bg_color=1, force_all_rays=False)
        loss = ((out["image"][0] - target) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    print(f"final held-out PSNR {log[-1][1]:.2f} dB after {args.steps} steps ({log[-1][2]:.1f} s, occupied cells {int((student.density_grid > 0.01).sum())})")
    return log


if __name__ == "__main__":
    main()
