#!/usr/bin/env python3
"""configs[3]-shaped training step (bench.make_training_step: PaletteNeRF or NeRF, 4096 rays/step, forward-facing slab scene,
dt_gamma = 1/128, Adam): ms/step of march_rays_train -> field -> composites -> backward -> optimiser step.  Synthetic data."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--model", choices=["palette", "nerf"], default="palette")
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--sync-each-step", action="store_true", help="loss.item()-style host read every step, as the reference's trainer does")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of palettenerf_amd.optim.Adam (pnr_adam_step: one launch for all tensors, same bits)")
    ap.add_argument("--torch-loss", action="store_true", help="the trainer's loss written with torch on the result dict instead of palettenerf_amd.train_loss")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m, step = bench.make_training_step(args.model, args.rays, dev, fp16=args.fp16, torch_adam=args.torch_adam, torch_loss=args.torch_loss)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        if args.sync_each_step:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    samples = int(m.step_counter[(m.local_step - 1) % 16, 0])
    print(f"{args.model} train{' fp16' if args.fp16 else ''}: {dt / args.steps * 1e3:.2f} ms/step, {samples} samples/step, {samples * args.steps / dt / 1e6:.1f} M samples/s")


if __name__ == "__main__":
    main()
