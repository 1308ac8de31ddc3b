#!/usr/bin/env python3
"""One frame as S interleaved-tile shards (the multi-GPU split of dist.py) rendered CONCURRENTLY on one GPU: S host threads, fused-field objects and
streams (pipeline.FramesInFlight).  Latency of a frame against the single-schedule frame.  usage: split_frame_in_flight.py [--workload lego] [--steps 30]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import dist as pdist  # noqa: E402
from palettenerf_amd.fused import tile_ray_order  # noqa: E402
from palettenerf_amd.pipeline import FramesInFlight  # noqa: E402

args = bench.parse(sys.argv[1:] + ["--no-cpu-baseline"])
dev = torch.device("cuda", 0)
wl = args.wl
H, W = wl["H"], wl["W"]
kw = dict(perturb=False, dt_gamma=wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
if args.model == "palette":
    kw["gui_mode"] = False
steps = args.steps
m = bench.build_model(args, dev)
full_idx, _ = pdist.shard_indices(H, W, 0, 1)
full_bank = bench.RayBank(args, 1, full_idx, dev)
m._fused.ray_order = tile_ray_order(full_idx, W, 8).to(dev)
for S in (1, 2, 3, 4):
    fif = FramesInFlight(m, S, dev)
    banks = []
    for k in range(S):
        idx, _ = pdist.shard_indices(H, W, k, S)
        banks.append(bench.RayBank(args, 1, idx, dev))
        fif.models[k]._fused.ray_order = tile_ray_order(idx, W, 8).to(dev)
    for i in range(steps + 3):
        for b in banks:
            b.get(i)

    def frame(i):
        return fif.render(lambda k: banks[k].get(i), S, consume=lambda k, r: int(r["rendered"].sum()), **kw)

    for i in range(3):
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for i in range(3, 3 + steps):
        n += sum(frame(i))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.workload}: {S} shard(s) of a frame in flight: {dt / steps * 1e3:.3f} ms per frame ({n / dt / 1e9:.3f} G samples/s)")
m._fused.ray_order = tile_ray_order(full_idx, W, 8).to(dev)
