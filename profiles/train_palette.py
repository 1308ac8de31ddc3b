#!/usr/bin/env python3
"""BASELINE configs[3] end to end on synthetic data (no datasets offline): `-m palette` training, 4096 rays per step, forward-facing rig
(17 views, 1008 x 756, dt_gamma 1/128, min_near 0.02: scripts/configs_llff/fern.sh) -- the reference's two-stage recipe (scripts/run_llff.sh):
  stage 1  a vanilla NeRF learns geometry + colour from the images (`-m nerf`);
  stage 2  a PaletteNetwork is initialised from that checkpoint (encoder / sigma_net / color_net load by name, palette/network.py:58-59) and
           from an "extracted" palette (here: the teacher's basis colours -- palette extraction itself is out of scope), then trained for
           --steps iterations with PaletteTrainer.train_step's loss (palette/utils.py:449-581: MSE + direct-colour MSE + lambda_sparsity,
           lambda_offsets, lambda_view_dep, lambda_palette terms), geometry frozen (sigma detached), Adam(lr 1e-2, betas (0.9, 0.99), eps 1e-15).
The teacher is a seeded PaletteNetwork with a smooth table; ground-truth images are rendered with the native frame loop.  Prints wall time,
ms/step and held-out PSNR; --optimizer torch runs torch.optim.Adam instead of the one-launch pnr_adam_step (same bits, more launches)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from palettenerf_amd import checkpoint, network, optim, raymarching, rays, renderer, scene  # noqa: E402
from palettenerf_amd.train_loss import train_loss  # noqa: E402


def rig(n, dev):
    poses = []
    for i in range(n):  # forward-facing rig: cameras on a 0.3-radius disc at z = 1.5 looking down -z
        a = 2 * np.pi * i / n
        p = np.eye(4, dtype=np.float32)
        p[:3, 0], p[:3, 1], p[:3, 2] = [1, 0, 0], [0, -1, 0], [0, 0, -1]
        p[:3, 3] = [0.3 * np.cos(a), 0.3 * np.sin(a), 1.5]
        poses.append(p)
    return torch.from_numpy(np.stack(poses)).to(dev)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5000, help="stage-2 (PaletteNeRF) iterations: configs[3] says 5k")
    ap.add_argument("--nerf-steps", type=int, default=3000, help="stage-1 (vanilla NeRF) iterations")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--res", type=float, default=0.25, help="fraction of 1008 x 756 (memory / teacher render time only; the step does not depend on it)")
    ap.add_argument("--optimizer", choices=["pnr", "torch"], default="pnr")
    ap.add_argument("--log-every", type=int, default=500)
    ap.add_argument("--lr-schedule", choices=["reference", "constant"], default="reference",
                    help="reference: LambdaLR(0.1 ** min(iter / iters, 1)) stepped every iteration, as main_palette.py:225-228 / main_nerf.py build it; constant: lr 1e-2 "
                         "throughout (rounds 1-4 of this script: the held-out PSNR then wanders by +-2 dB from 2 000 steps on and fell 40.9 -> 36.9 dB between steps "
                         "4 000 and 5 000 of round 3's run -- Adam at lr 1e-2 on 50 MB hash tables does not settle without the decay)")
    ap.add_argument("--dead-rows", action="store_true", help="after training: how many samples of a step get an all-zero gradient from the composites (profiles/dead_rows.py)")
    ap.add_argument("--torch-loss", action="store_true", help="the losses written with torch on the result dict instead of palettenerf_amd.train_loss")
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    W, H = int(1008 * args.res), int(756 * args.res)
    intr = scene.intrinsics_from_fov(H, W, 0.9)
    poses = rig(18, dev)           # 17 training views + 1 held out (fern: 20 images, every 8th held out)
    kw = dict(dt_gamma=1.0 / 128, max_steps=1024, T_thresh=1e-4, bg_color=1)
    make_opt = (lambda p: optim.Adam(p, betas=(0.9, 0.99), eps=1e-15)) if args.optimizer == "pnr" else (lambda p: torch.optim.Adam(p, betas=(0.9, 0.99), eps=1e-15))

    # ---------------- teacher + ground truth
    opt_ns = renderer.default_opt(test=False)
    teacher = network.PaletteNetwork(opt_ns, bound=2, cuda_ray=True, density_scale=30.0, min_near=0.02)
    scene.seed_field_(teacher, 0)
    with torch.no_grad():   # smooth teacher: only the coarse levels of its tables are non-zero
        off = teacher.encoder.offsets
        for enc in (teacher.encoder, teacher.encoder_palette):
            enc.embeddings[int(off[6]):] = 0
    teacher = teacher.to(dev).eval()
    teacher.density_grid.copy_(torch.from_numpy(scene.slab_density_grid()).to(dev))
    raymarching.packbits(teacher.density_grid, 0.5, teacher.density_bitfield)
    teacher.march_mode, teacher.fused_field = "native", True
    images = []
    with torch.no_grad():
        for p in poses:
            r = rays.get_rays(p[None], intr, H, W, -1)
            images.append(teacher.render(r["rays_o"], r["rays_d"], perturb=False, gui_mode=True, **kw)["image"][0])
    images = torch.stack(images)
    palette = teacher.basis_color.detach().clamp(0, 1).cpu().tolist()
    print(f"teacher: 17 training views + 1 held-out, {W}x{H}, mean colour {[round(v, 3) for v in images.mean(dim=(0, 1)).tolist()]}")

    def evaluate(m, **extra):
        m.eval()
        m.march_mode, m.fused_field = "native", True
        with torch.no_grad():
            r = rays.get_rays(poses[-1:], intr, H, W, -1)
            img = m.render(r["rays_o"], r["rays_d"], perturb=False, **kw, **extra)["image"][0]
        m.train()
        return scene.psnr(img, images[-1])

    def batch():
        v = int(torch.randint(0, 17, (1,)))
        r = rays.get_rays(poses[v:v + 1], intr, H, W, args.rays)
        return r["rays_o"], r["rays_d"], images[v][r["inds"][0]][None]

    # ---------------- stage 1: vanilla NeRF (geometry)
    nerf = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=1.0, min_near=0.02).to(dev).train()
    o1 = make_opt(nerf.get_params(1e-2))
    sched1 = torch.optim.lr_scheduler.LambdaLR(o1, lambda it: 0.1 ** min(it / max(1, args.nerf_steps), 1)) if args.lr_schedule == "reference" else None   # main_nerf.py:147
    t0 = time.perf_counter()
    for step in range(args.nerf_steps):
        if step % 16 == 0:
            with torch.no_grad():
                nerf.update_extra_state()
        ro, rd, gt = batch()
        out = nerf.render(ro, rd, perturb=True, force_all_rays=False, **kw)
        loss = ((out["image"] - gt) ** 2).mean() if args.torch_loss else train_loss(out, gt)[0]
        o1.zero_grad(set_to_none=True)
        loss.backward()
        o1.step()
        if sched1 is not None:
            sched1.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    print(f"stage 1 (-m nerf): {args.nerf_steps} steps in {t1:.1f} s ({t1 / max(1, args.nerf_steps) * 1e3:.2f} ms/step incl. occupancy updates), held-out PSNR {evaluate(nerf):.2f} dB")
    ckpt = checkpoint.save_model(nerf, "/tmp/pnr_stage1.pth", epoch=1, global_step=args.nerf_steps)

    # ---------------- stage 2: PaletteNeRF from the NeRF checkpoint + the extracted palette
    pal = network.PaletteNetwork(opt_ns, bound=2, cuda_ray=True, density_scale=1.0, min_near=0.02).to(dev)
    info = checkpoint.load_model(pal, ckpt, map_location=dev)
    pal.initialize_palette(palette)
    pal.to(dev).train()
    print(f"stage 2 initialised from the stage-1 checkpoint: {len(info['missing'])} palette-only entries start fresh, unexpected {info['unexpected']}")
    o2 = make_opt(pal.get_params(1e-2))
    sched2 = torch.optim.lr_scheduler.LambdaLR(o2, lambda it: 0.1 ** min(it / max(1, args.steps), 1)) if args.lr_schedule == "reference" else None
    lam = dict(sparsity=2e-4, offsets=0.03, view_dep=0.1, palette=0.001)   # main_palette.py:83-89
    log = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(args.steps + 1):
        if step % args.log_every == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            t_eval = time.perf_counter()
            log.append((step, evaluate(pal, gui_mode=True), dt))
            print(f"step {step:5d}  held-out PSNR {log[-1][1]:6.2f} dB   {dt:6.1f} s training wall")
            torch.cuda.synchronize()
            t0 += time.perf_counter() - t_eval       # evaluation time is not training time
            if step == args.steps:
                break
        ro, rd, gt = batch()
        out = pal.render(ro, rd, perturb=True, force_all_rays=True, **kw)
        if not args.torch_loss:    # PaletteTrainer.train_step's loss in one launch each way (palettenerf_amd.train_loss)
            loss, _ = train_loss(out, gt, lambda_sparsity=lam["sparsity"], lambda_offsets=lam["offsets"], lambda_view_dep=lam["view_dep"], lambda_palette=lam["palette"],
                                 basis_color=pal.basis_color, basis_color_origin=pal.basis_color_origin, want_outputs=False)
            o2.zero_grad(set_to_none=True)
            loss.backward()
            o2.step()
            if sched2 is not None:
                sched2.step()
            continue
        loss = ((out["image"] - gt) ** 2).mean(-1)
        loss = loss + lam["sparsity"] * out["omega_sparsity"].mean() + lam["offsets"] * out["offsets_norm"].mean() + lam["view_dep"] * out["view_dep_norm"].mean()
        loss = loss + lam["palette"] * ((pal.basis_color - pal.basis_color_origin) ** 2).sum(dim=-1).mean() + ((out["direct_rgb"] - gt) ** 2).mean()
        loss = loss.mean()
        o2.zero_grad(set_to_none=True)
        loss.backward()
        o2.step()
        if sched2 is not None:
            sched2.step()
    total = log[-1][2]
    samples = float(pal.step_counter[:, 0].float().mean())      # the last 16 steps' sample counts
    print(f"stage 2 (-m palette, configs[3]): {args.steps} steps in {total:.1f} s = {total / args.steps * 1e3:.2f} ms/step (optimizer: {args.optimizer}), "
          f"final held-out PSNR {log[-1][1]:.2f} dB; {samples / 1e6:.2f} M samples per step at the end ({total / args.steps * 1e3 / (samples / 1e6):.2f} ms per M samples)")
    if args.dead_rows:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import dead_rows

        def one_step(_i):
            ro, rd, gt = batch()
            out = pal.render(ro, rd, perturb=True, force_all_rays=True, **kw)
            loss, _ = train_loss(out, gt, lambda_sparsity=lam["sparsity"], lambda_offsets=lam["offsets"], lambda_view_dep=lam["view_dep"], lambda_palette=lam["palette"],
                                 basis_color=pal.basis_color, basis_color_origin=pal.basis_color_origin, want_outputs=False)
            o2.zero_grad(set_to_none=True)
            loss.backward()
        dead_rows.probe("trained scene (palette)", pal, one_step)
    try:   # device time of a step's kernels at the end of training (the wall figure above includes the host's share)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                ro, rd, gt = batch()
                out = pal.render(ro, rd, perturb=True, force_all_rays=True, **kw)
                loss, _ = train_loss(out, gt, lambda_sparsity=lam["sparsity"], lambda_offsets=lam["offsets"], lambda_view_dep=lam["view_dep"], lambda_palette=lam["palette"],
                                     basis_color=pal.basis_color, basis_color_origin=pal.basis_color_origin, want_outputs=False)
                o2.zero_grad(set_to_none=True)
                loss.backward()
                o2.step()
            torch.cuda.synchronize()
        ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "memcpy" not in e.name.lower() and "memset" not in e.name.lower()]
        print(f"          10 more steps under torch.profiler: {sum(e.device_time for e in ks) / 10 / 1e3:.2f} ms of kernels and {len(ks) / 10:.0f} launches per step")
    except Exception as e:   # noqa: BLE001
        print("          profiler:", repr(e))
    return log


if __name__ == "__main__":
    main()
