#!/usr/bin/env python3
"""configs[3]-shaped training step (PaletteNeRF, 4096 rays/step, forward-facing slab scene, dt_gamma = 1/128, Adam):
ms/step of march_rays_train -> field -> composite_rays_train + composite_rays_flex_train -> backward (composite bwd,
MLP bwd, grid_encode backward with hardware atomics) -> optimiser step.  Synthetic data (no datasets offline)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from palettenerf_amd import network, raymarching, renderer, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--model", choices=["palette", "nerf"], default="palette")
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--sync-each-step", action="store_true", help="loss.item() every step, as the reference's trainer does")
    ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): one launch per parameter group instead of seven")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of palettenerf_amd.optim.Adam (pnr_adam_step: one launch for all tensors, same bits)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.model == "palette":
        m = network.PaletteNetwork(renderer.default_opt(test=False), bound=2, cuda_ray=True, min_near=0.02)
    else:
        m = network.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.02)
    scene.seed_field_(m, 0)
    m = m.to(dev).train()
    m.density_grid.copy_(torch.from_numpy(scene.slab_density_grid()).to(dev))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    H, W = 756, 1008
    g = torch.Generator().manual_seed(0)
    poses = []
    for i in range(17):  # forward-facing rig: cameras on a 0.3-radius disc at z = 1.5 looking down -z
        a = 2 * np.pi * i / 17
        p = np.eye(4, dtype=np.float32)
        p[:3, 0], p[:3, 1], p[:3, 2] = [1, 0, 0], [0, -1, 0], [0, 0, -1]
        p[:3, 3] = [0.3 * np.cos(a), 0.3 * np.sin(a), 1.5]
        poses.append(p)
    intr = scene.intrinsics_from_fov(H, W, 0.9)
    ro_all, rd_all = scene.get_rays(torch.from_numpy(np.stack(poses)), intr, H, W)
    ro_all, rd_all = ro_all.to(dev), rd_all.to(dev)
    if args.torch_adam or args.fused_adam or args.fp16:
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=args.fused_adam)
    else:
        from palettenerf_amd import optim
        opt = optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda", enabled=args.fp16)
    target = torch.rand(args.rays, 3, device=dev)

    def step(i):
        inds = torch.randint(0, H * W, [args.rays], generator=g).to(dev)
        ro, rd = ro_all[i % 17, inds][None], rd_all[i % 17, inds][None]
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.fp16):
            r = m.run_cuda(ro, rd, dt_gamma=1 / 128, perturb=True, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
            loss = ((r["image"][0] - target) ** 2).mean()
            if args.model == "palette":
                loss = loss + 1e-3 * r["omega_sparsity"].mean() + 1e-2 * r["offsets_norm"].mean() + ((r["direct_rgb"][0] - target) ** 2).mean()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        if args.sync_each_step:
            loss.item()      # the reference's trainer reads the loss every step (nerf/utils.py train_one_epoch): host and GPU cannot overlap across steps
        return int(m.step_counter[(m.local_step - 1) % 16, 0])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = 0
    for i in range(args.steps):
        samples += step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.model} train{' fp16' if args.fp16 else ''}: {dt / args.steps * 1e3:.2f} ms/step, {samples / args.steps:.0f} samples/step, "
          f"{samples / dt / 1e6:.1f} M samples/s")


if __name__ == "__main__":
    main()
