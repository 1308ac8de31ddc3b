#!/usr/bin/env python3
"""Iterations and host looks (stream synchronisations inside the frame call) per frame along the bench's camera path: how often the chunk the frame
loop enqueues from the previous frame's iteration count falls short.  usage: python profiles/frame_looks.py [--workload lego] [--frames 40]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import dist as pdist  # noqa: E402
from palettenerf_amd.fused import tile_ray_order  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="lego")
ap.add_argument("--frames", type=int, default=40)
a = ap.parse_args()
args = bench.parse(["--workload", a.workload, "--no-cpu-baseline"])
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
H, W = args.wl["H"], args.wl["W"]
idx, _ = pdist.shard_indices(H, W, 0, 1)
bank = bench.RayBank(args, 1, idx, dev)
m._fused.ray_order = tile_ray_order(idx, W, 8).to(dev)
kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
if args.model == "palette":
    kw["gui_mode"] = False
its, looks = [], []
with torch.no_grad():
    for i in range(a.frames):
        r = m.render(*bank.get(i), **kw)
        its.append(r["iterations"]); looks.append(r["host_looks"])
print(a.workload, "iterations per frame:", its)
print(a.workload, "looks per frame:     ", looks, f"-> {sum(1 for x in looks[1:] if x > 1)} of {len(looks) - 1} frames needed more than one look")
